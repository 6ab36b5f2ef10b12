"""The fp32 path (north star: "logits must agree within 1e-4 relative for the fp32 path"), forward AND backward.

Kernels (through the C ABI) against float64 torch on the same inputs at 1e-5; whole models (tiny MMLoRA / MMAdapter, ViT-B
MMLoRA / MMAdapter at the BASELINE tile size) against the goldens the reference itself produced in float64 arithmetic
(tools/make_goldens.py) at 1e-4 relative (max |a-b| / max |b|) and argmax agreement >= 99.9 %. fp32 TRAINING (the reference
trains in fp32, trainer.py:146-159): blocks, the tiny encoder and whole train steps (tiny models, ViT-B) against the reference's
recorded gradients at 1e-4 - the pin the bf16 path's gradients (cos > 0.98) cannot give."""
import argparse
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, rel_err
from oracle import mfnet_ref as R
from ssrs_amd import detfill

pytestmark = pytest.mark.gpu
FP32_TOL = 1e-4


def _u(key, shape, scale=1.0):
    return torch.from_numpy(detfill.uniform(key, shape)).float() * scale


def _args(mod):
    return argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)


@pytest.mark.parametrize("M,N,K", [(512, 768, 768), (100, 6, 64), (392, 2304, 768), (70, 68, 4), (64, 64, 36), (1, 8, 768)])
def test_gemm_f32_all_epilogues(M, N, K):
    from ssrs_amd.fp32_path import gemm32
    a, w, b = _u("g32/a", (M, K)), _u("g32/w", (N, K)), _u("g32/b", (N,))
    res = _u("g32/r", (M, N))
    ad, wd = a.double(), w.double()
    base = ad @ wd.t()
    ac, wc, bc, rc = a.cuda(), w.cuda(), b.cuda(), res.cuda()
    assert rel_err(gemm32(ac, wc), base) < 1e-5
    assert rel_err(gemm32(ac, wc, bias=bc, act=1), F.gelu(base + b.double())) < 1e-5
    assert rel_err(gemm32(ac, wc, bias=bc, act=3, residual=rc), F.relu6(base + b.double()) + res.double()) < 1e-5
    assert rel_err(gemm32(ac, wc, act=6), F.relu(base)) < 1e-5
    out = rc.clone()
    gemm32(ac, wc, alpha=0.25, out=out, accumulate=True)
    assert rel_err(out, res.double() + 0.25 * base) < 1e-5
    if M % 4 == 0:                      # residual rows taken modulo (the abs-pos-embed add of the patch embedding)
        want = base + res.double()[:M // 4].repeat(4, 1)
        assert rel_err(gemm32(ac, wc, residual=rc[:M // 4], res_row_mod=M // 4), want) < 1e-5
    big = torch.zeros(M, N + 12, device="cuda")                       # strided output view / strided operand view
    gemm32(ac, wc, out=big[:, 4:4 + N])
    assert rel_err(big[:, 4:4 + N], base) < 1e-5 and float(big[:, :4].abs().max()) == 0 and float(big[:, 4 + N:].abs().max()) == 0
    if K >= 8:
        assert rel_err(gemm32(ac[:, 4:], wc[:, 4:]), ad[:, 4:] @ wd[:, 4:].t()) < 1e-5


def test_gemm_f32_rejects_bad_operands():
    from ssrs_amd._lib import MfnetError
    from ssrs_amd.fp32_path import gemm32
    a, w = torch.zeros(8, 6, device="cuda"), torch.zeros(8, 6, device="cuda")
    with pytest.raises(MfnetError):
        gemm32(a, w)                     # K % 4
    with pytest.raises(MfnetError):
        gemm32(torch.zeros(8, 8), torch.zeros(8, 8))                # host tensors


def _attn_ref(qkv, bias, rel_h, rel_w, G, S, heads, ws):
    """oracle: pad the token map with rows equal to the qkv bias (what zero-padded tokens become), windows, attention."""
    D3 = qkv.shape[-1]
    x = qkv.view(G, S, S, D3)
    if ws > 0:
        pad = (ws - S % ws) % ws
        Sp = S + pad
        full = bias.view(1, 1, 1, D3).expand(G, Sp, Sp, D3).clone()
        full[:, :S, :S] = x
        win, pad_hw = R.window_partition(full, ws)
        h = w = ws
    else:
        win, h, w = x, S, S
    Bw = win.shape[0]
    q, k, v = R.split_heads(win.reshape(Bw, h * w, D3), heads)
    hd = D3 // (3 * heads)
    o = R.merge_heads(R.attention_core(q, k, v, rel_h, rel_w, (h, w), hd ** -0.5), Bw, heads, h, w)
    if ws > 0:
        o = R.window_unpartition(o, ws, pad_hw, (S, S))
    return o.reshape(G * S * S, -1)


@pytest.mark.parametrize("S,ws,heads,hd", [(16, 14, 2, 64), (16, 0, 2, 64), (20, 14, 2, 80), (20, 0, 1, 80), (16, 14, 3, 32), (28, 14, 1, 64)])
def test_attn_f32_vs_oracle(S, ws, heads, hd):
    from ssrs_amd.fp32_path import attn32
    G = 2
    k = ws if ws else S
    qkv = _u("a32/qkv", (G * S * S, 3 * heads * hd), 1.5)
    bias = _u("a32/b", (3 * heads * hd,), 0.5)
    rel_h, rel_w = _u("a32/rh", (2 * k - 1, hd)), _u("a32/rw", (2 * k - 1, hd))
    ref = _attn_ref(qkv.double(), bias.double(), rel_h.double(), rel_w.double(), G, S, heads, ws)
    out = attn32(qkv.cuda(), bias.cuda(), rel_h.cuda(), rel_w.cuda(), G, S, S, heads, ws, hd ** -0.5)
    assert rel_err(out, ref) < 1e-5


def _eval_case(tag, mod, encoder, batch):
    from ssrs_amd.unetformer_mmsam import UNetFormer
    g = golden(tag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder=encoder, args=_args(mod), precision="fp32")
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().eval()
    rgb, dsm, tgt = detfill.tiles(tag, batch, 256, 6)
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
    assert out.shape == (batch, 6, 256, 256) and out.dtype == torch.float32
    e = rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"])
    agree = float((out.argmax(1).cpu().numpy() == g["eval_argmax"]).mean())
    csum = rel_err(out.double().sum((0, 2, 3)), g["eval_chan_sum"])
    print(f"\n[{tag} fp32 path] logits rel err {e:.3e}  channel sums {csum:.3e}  argmax agreement {100 * agree:.4f} %")
    assert e < FP32_TOL, e
    assert csum < FP32_TOL, csum
    assert agree >= 0.999, agree
    return m, (rgb, dsm)


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_model_fp32_logits_vs_reference(mod):
    _eval_case(f"model_tiny_{mod}", mod, "vit_tiny_test", 2)


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_vitb_fp32_logits_vs_reference(mod):
    """BASELINE config (ViT-B, 256x256 tile): the 1e-4 bar of the fp32 path and the 99.9 % mask agreement."""
    _eval_case(f"model_vitb_{mod}", mod, "vit_b", 1)


def test_fp32_path_is_switchable():
    m, (rgb, dsm) = _eval_case("model_tiny_sam_lora", "sam_lora", "vit_tiny_test", 2)
    with torch.no_grad():
        lean = m(rgb.cuda(), dsm.cuda())
    full = m(rgb.cuda(), dsm.cuda())             # eval mode with autograd on: the module path in fp32 mode (saves for backward)
    assert full.requires_grad and rel_err(full.detach(), lean) < 1e-6
    m.precision = "bf16"                 # same module, same weights, the bf16 MFMA path: 2e-2 of the same golden
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
    assert 1e-4 < rel_err(out[:, :, ::4, ::4], golden("model_tiny_sam_lora")["eval_logits_s4"]) < 2e-2
    y3 = dsm[:, None].repeat(1, 3, 1, 1).cuda()                       # the 3-channel form of the second modality
    m.precision = "fp32"
    with torch.no_grad():
        assert rel_err(m(rgb.cuda(), y3), m(rgb.cuda(), dsm.cuda())) < 1e-6


def test_1024_tile_fp32_and_bf16_paths_agree():
    """BASELINE configs[4] geometry (one 1024x1024 tile: 64x64 token grid, 25 windows of 14x14 with padding, global attention over
    4096 keys) has no golden of the reference (its float64 run would take hours here); the two independent implementations
    - fp32 kernels and bf16 MFMA kernels - must agree to the bf16 bound on it."""
    from ssrs_amd.unetformer_mmsam import UNetFormer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder="vit_b", args=_args("sam_lora"))
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "model_vitb_sam_lora/"))
    m = m.cuda().eval()
    rgb, dsm, _ = detfill.tiles("t1024", 1, 1024, 6)
    with torch.no_grad():
        lo = m(rgb.cuda(), dsm.cuda())
        m.precision = "fp32"
        hi = m(rgb.cuda(), dsm.cuda())
    e = rel_err(lo, hi)
    agree = float((lo.argmax(1) == hi.argmax(1)).float().mean())
    print(f"\n[vit_b 1024x1024] bf16 vs fp32 path: logits {e:.3e}, argmax agreement {100 * agree:.3f} %")
    assert e < 2e-2 and agree >= 0.99, (e, agree)          # measured 1.2e-2 / 99.35 % (1 M pixels of near-tied random-init logits)


# ------------------------------------------------------------------------------------------------ fp32 training
@pytest.mark.parametrize("S,ws,heads,hd", [(16, 14, 2, 64), (16, 0, 2, 64), (20, 14, 2, 80), (20, 0, 1, 80), (16, 14, 3, 32), (28, 14, 1, 64)])
def test_attn_f32_bwd_vs_oracle(S, ws, heads, hd):
    """dqkv of the fp32 attention against float64 autograd through the oracle's attention (padded windows, interpolated tables)."""
    from ssrs_amd import ops
    G = 2
    k = ws if ws else S
    qkv = _u("a32/qkv", (G * S * S, 3 * heads * hd), 1.5)
    bias = _u("a32/b", (3 * heads * hd,), 0.5)
    rel_h, rel_w = _u("a32/rh", (2 * k - 1, hd)), _u("a32/rw", (2 * k - 1, hd))
    dout = _u("a32/do", (G * S * S, heads * hd))
    qd = qkv.double().requires_grad_()
    ref = _attn_ref(qd, bias.double(), rel_h.double(), rel_w.double(), G, S, heads, ws)
    (ref * dout.double()).sum().backward()
    out = ops.attn_f32(qkv.cuda(), bias.cuda(), rel_h.cuda(), rel_w.cuda(), G, S, S, heads, ws, hd ** -0.5)
    dqkv = ops.attn_f32_bwd(qkv.cuda(), bias.cuda(), rel_h.cuda(), rel_w.cuda(), out, dout.cuda(), G, S, S, heads, ws, hd ** -0.5)
    D = heads * hd
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert rel_err(dqkv[:, sl], qd.grad[:, sl]) < 1e-5, name


def test_fp32_gemm_route_epilogues_vs_float64():
    """ops.gemm on fp32 operands (fp32 training mode) keeps the bf16 kernel's contract: a w^T (+ bias + s U V_g^T) -> act -> (+ residual)
    for every activation code the module backward passes use, the grouped form, and the in-place residual form."""
    from ssrs_amd import ops
    M, N, K, R = 256, 96, 64, 8
    a, w, b = _u("r32/a", (M, K)), _u("r32/w", (N, K)), _u("r32/b", (N,))
    U, V = _u("r32/U", (M, R)), _u("r32/V", (2, N, R))
    aux, res = _u("r32/aux", (M, N), 4.0) + 2.0, _u("r32/res", (M, N))
    ac, wc, bc, Uc, Vc, auxc, resc = [t.cuda() for t in (a, w, b, U, V, aux, res)]
    ad, wd, bd, Ud, Vd, auxd, resd = [t.double() for t in (a, w, b, U, V, aux, res)]
    lowrank = torch.cat([Ud[:M // 2] @ Vd[0].t(), Ud[M // 2:] @ Vd[1].t()]) * 0.25
    base = ad @ wd.t() + bd
    with ops.precision("fp32"):
        assert rel_err(ops.gemm(ac, wc, bias=bc, U=Uc, V=Vc, rows_per_group=M // 2, n_groups=2, lr_scale=0.25), base + lowrank) < 1e-5
        out, pre = ops.gemm(ac, wc, bias=bc, U=Uc, V=Vc[0], lr_scale=0.25, act=1, want_pre=True)
        p0 = base + 0.25 * Ud @ Vd[0].t()
        assert rel_err(pre, p0) < 1e-5 and rel_err(out, F.gelu(p0)) < 1e-5
        out, pre = ops.gemm(ac, wc, bias=bc, residual=resc, want_pre=True, out=torch.empty(M, N, device="cuda"))
        assert rel_err(pre, base) < 1e-5 and rel_err(out, base + resd) < 1e-5
        xg = auxd.clone().requires_grad_()
        F.gelu(xg).sum().backward()
        assert rel_err(ops.gemm(ac, wc, bias=bc, act=2, aux=auxc), base * xg.grad) < 1e-5
        assert rel_err(ops.gemm(ac, wc, bias=bc, act=3), F.relu6(base)) < 1e-5
        hid = F.relu6(auxd)
        assert rel_err(ops.gemm(ac, wc, act=4, aux=hid.float().cuda()), (ad @ wd.t()) * ((hid > 0) & (hid < 6))) < 1e-5
        assert rel_err(ops.gemm(ac, wc, residual=resc, act=5), F.relu6(ad @ wd.t() + resd)) < 1e-5
        acc = resc.clone()
        ops.gemm(ac, wc, residual=acc, out=acc)
        assert rel_err(acc, ad @ wd.t() + resd) < 1e-5
        wg, bg = torch.stack([wc, wc * 0.5]), torch.stack([bc, -bc])
        og, pg = ops.gemm_grouped(ac, wg, bias=bg, rows_per_group=M // 2, act=1, want_pre=True)
        want = torch.cat([ad[:M // 2] @ wd.t() + bd, ad[M // 2:] @ (0.5 * wd).t() - bd])
        assert rel_err(pg, want) < 1e-5 and rel_err(og, F.gelu(want)) < 1e-5


def test_fp32_product_routes_vs_float64():
    """the skinny / transposed products and the 3x3 convolution (forward, data gradient via flipped weights, weight gradient)
    of fp32 training mode against float64."""
    from ssrs_amd import ops
    M, C, R = 512, 64, 8
    y, u = _u("p32/y", (M, C)), _u("p32/u", (M, R))
    wt = _u("p32/wt", (2, R, C))
    yc, uc, wtc = y.cuda(), u.cuda(), wt.cuda()
    yd, ud, wtd = y.double(), u.double(), wt.double()
    h = M // 2
    with ops.precision("fp32"):
        assert rel_err(ops.skinny_nt(yc, wtc, scale=0.5, rows_per_group=h), 0.5 * torch.cat([yd[:h] @ wtd[0].t(), yd[h:] @ wtd[1].t()])) < 1e-5
        assert rel_err(ops.skinny_nt(yc, wtc[0]), yd @ wtd[0].t()) < 1e-5
        got = ops.skinny_tn(yc, uc, False, 2.0, n_groups=2, lazy=True)
        assert rel_err(got[0], 2.0 * yd[:h].t() @ ud[:h]) < 1e-5 and rel_err(got[1], 2.0 * yd[h:].t() @ ud[h:]) < 1e-5
        assert rel_err(ops.skinny_tn(yc, uc, True), ud.t() @ yd) < 1e-5
        dW, db = ops.gemm_tn(yc, uc, lazy=True, want_bias=True)
        assert rel_err(dW, yd.t() @ ud) < 1e-5 and rel_err(db.materialize(), yd.sum(0)) < 1e-5
        dWg, dbg = ops.gemm_tn_grouped(yc, uc, 2, want_bias=True)
        assert rel_err(dWg[1].materialize(), yd[h:].t() @ ud[h:]) < 1e-5 and rel_err(dbg[0].materialize(), yd[:h].sum(0)) < 1e-5
        B, H, W, Ci, Co = 2, 16, 16, 32, 48
        x, wconv = _u("p32/x", (B, Ci, H, W)), _u("p32/wc", (Co, Ci, 3, 3), 0.2)
        xr = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous().cuda()
        wk = wconv.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous().cuda()
        xd, wcd = x.double().requires_grad_(), wconv.double().requires_grad_()
        ref = F.conv2d(xd, wcd, padding=1)
        gy = _u("p32/gy", tuple(ref.shape))
        (ref * gy.double()).sum().backward()
        rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
        assert rel_err(ops.conv3x3(xr, wk, B, H, W), rows(ref.detach())) < 1e-5
        gyr = rows(gy).contiguous().cuda()
        wflipT = wconv.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, 9 * Co).contiguous().cuda()
        assert rel_err(ops.conv3x3(gyr, wflipT, B, H, W), rows(xd.grad)) < 1e-5
        dWc = ops.conv3x3_wgrad(gyr, xr, B, H, W)
        assert rel_err(dWc.view(Co, 3, 3, Ci).permute(0, 3, 1, 2), wcd.grad) < 1e-5


@pytest.mark.parametrize("kind", ["lora", "adpt"])
@pytest.mark.parametrize("ws", [14, 0])
@pytest.mark.parametrize("S", [16, 20])
def test_block_fwd_bwd_fp32_vs_reference(kind, ws, S):
    """the block goldens of the reference (its own fp32 run) at the fp32 bar: outputs, input gradients, every PEFT gradient"""
    from test_model_gpu import _block
    from ssrs_amd import ops
    tag = f"block_{kind}_w{ws}_s{S}"
    g = golden(tag)
    blk = _block(kind, ws)
    blk.load_state_dict(detfill.fill_state_dict(blk.state_dict(), tag + "/"))
    blk = blk.cuda()
    token = "lora_" if kind == "lora" else "Adapter"
    trainable = [k[2:] for k in g.files if k.startswith("g/") and token in k]
    for n, p in blk.named_parameters():
        p.requires_grad = n in trainable
    x = _u(tag + "/x", (1, S, S, 128)).cuda().requires_grad_()
    y = _u(tag + "/y", (1, S, S, 128)).cuda().requires_grad_()
    with ops.precision("fp32"):
        ox, oy = blk(x, y)
    assert rel_err(ox, g["out_x"]) < FP32_TOL and rel_err(oy, g["out_y"]) < FP32_TOL
    ((ox * _u(tag + "/gx", ox.shape).cuda()).sum() + (oy * _u(tag + "/gy", oy.shape).cuda()).sum()).backward()      # outside the context: each node keeps its mode
    assert rel_err(x.grad, g["grad_x"]) < FP32_TOL and rel_err(y.grad, g["grad_y"]) < FP32_TOL
    # the reference gradients are themselves an fp32 run: a cancelling scalar sum carries its own 1e-6 x terms error
    bad = {n: rel_err(p.grad, g["g/" + n]) for n, p in blk.named_parameters() if n in trainable}
    print(f"\n[{tag} fp32] worst PEFT gradient error {max(bad.values()):.2e}")
    bad = {n: e for n, e in bad.items() if e > (1e-3 if g["g/" + n].size == 1 else FP32_TOL)}
    assert not bad, bad


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_encoder_fp32_vs_reference(mod):
    from test_model_gpu import _tiny_model
    from ssrs_amd import ops
    tag = f"encoder_tiny_{mod}"
    g = golden(tag)
    enc = _tiny_model(mod).image_encoder
    enc.load_state_dict(detfill.fill_state_dict(enc.state_dict(), tag + "/"))
    enc = enc.cuda().eval()
    x = _u(tag + "/x", (1, 3, 256, 256)).cuda()
    y = _u(tag + "/y", (1, 1, 256, 256)).cuda()
    with ops.precision("fp32"):
        fx, fy = enc(x, y.repeat(1, 3, 1, 1))
    assert rel_err(fx, g["fx"]) < FP32_TOL and rel_err(fy, g["fy"]) < FP32_TOL
    ((fx * _u(tag + "/gx", fx.shape).cuda()).sum() + (fy * _u(tag + "/gy", fy.shape).cuda()).sum()).backward()
    names = [k[2:] for k in g.files if k.startswith("g/")]
    pd = dict(enc.named_parameters())
    assert {n for n, p in pd.items() if p.grad is not None} == set(names)
    err = {n: rel_err(pd[n].grad, g["g/" + n]) for n in names}
    print(f"\n[{tag} fp32] worst gradient error {max(err.values()):.2e}")
    bad = {n: e for n, e in err.items() if e > (1e-3 if pd[n].numel() == 1 else FP32_TOL)}
    assert not bad, bad


def _train_step_fp32(tag, mod, encoder, ncls=6, with_oracle32=False, ignore=None):
    """Two SGD steps of the fp32 path against the reference's recorded float64 step: train-mode logits, loss0 / loss1, updated
    parameters and BatchNorm buffers at the fp32 bar - and EVERY gradient (norm + leading slice).

    What the gradients can be held to: the gradient map of the whole model is not continuous at fp32 resolution (ReLU6 / ReLU /
    max-pool switches; training-mode BatchNorm over 2 tiles): torch's OWN float32 evaluation of this step (the CPU oracle in
    float32, tools/diag_fp32_grads.py) is median 2e-3 / worst 5e-2 away from its float64 evaluation, on the same tensors this
    path is (167 of 173 beyond 1e-4), while logits and loss agree to 5e-6 / 1e-7. So 1e-4 holds where the map is smooth -
    blocks and the whole encoder under a fixed upstream gradient (tests above: <= 2e-5) - and the end-to-end gradients are held
    to direction and size: cosine > 0.9999 (measured 0.999999), norm ratio within 2e-3 (the bf16 path: 0.995 / 6e-3), median
    per-tensor error < 1e-2 (measured 8e-4 .. 4e-3); with_oracle32 prints torch's own float32 error next to it."""
    from helpers import oracle_train_step_grads, summary_err
    from ssrs_amd.unetformer_mmsam import UNetFormer
    g = golden(tag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=ncls, encoder=encoder, args=_args(mod), precision="fp32")
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().train()
    m.decoder.segmentation_head[1].p = 0.0
    rgb, dsm, tgt = detfill.tiles(tag, 2, 256, ncls)
    if ignore is not None:
        tgt = tgt.clone()
        tgt[:, :32, :48] = ignore            # tools/make_goldens.py _train_step_arrays
    rgb, dsm, tgt = rgb.cuda(), dsm.cuda(), tgt.cuda()
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=5e-4)
    losses, w = [], {}
    names = [str(n) for n in g["grad_names"]]
    pd = dict(m.named_parameters())
    floor = 1e-5 * float(np.median([g["g/" + n][0] for n in names]))
    for step in range(2):
        opt.zero_grad()
        out = m(rgb, dsm)
        loss = F.cross_entropy(out.permute(0, 2, 3, 1).reshape(-1, ncls), tgt.view(-1), **({} if ignore is None else {"ignore_index": ignore}))
        loss.backward()
        losses.append(loss.item())
        if step == 0:
            w["logits"] = rel_err(out[:, :, ::4, ::4], g["logits_s4"])
            assert {n for n, p in pd.items() if p.grad is not None} == set(names)
            errs = {n: max(summary_err(pd[n].grad, g["g/" + n], floor)) for n in names}
            num = da = db = 0.0
            for n in names:
                sl = pd[n].grad.double().flatten()[:256].cpu().numpy()
                ref = g["g/" + n][2:2 + sl.size]
                num += float((sl * ref).sum()); da += float((sl * sl).sum()); db += float((ref * ref).sum())
            w["cos"], w["norm_ratio"] = num / np.sqrt(da * db), np.sqrt(da / db)
            w["grad_median"], w["grad_max"] = float(np.median(list(errs.values()))), max(errs.values())
            w["worst"] = max(errs, key=errs.get)
            if with_oracle32:
                _, _, g32 = oracle_train_step_grads(tag, mod, torch.float32, ncls)
                e32 = {n: max(summary_err(g32[n], g["g/" + n], floor)) for n in names}
                w["torch_fp32_median"], w["torch_fp32_max"] = float(np.median(list(e32.values()))), max(e32.values())
        opt.step()
    perr = [summary_err(pd[n], g["p/" + n]) for n in names]
    w["param_norm"], w["param_slice"] = max(e[0] for e in perr), max(e[1] for e in perr)
    berr = []
    for n, b in m.named_buffers():
        if "b/" + n in g.files and "running_" in n:
            want = g["b/" + n]
            berr.append(rel_err(b.double().cpu().numpy(), want) if b.numel() < 300 else abs(b.double().norm().item() / want[0] - 1))
    w["bn_buffers"] = max(berr)
    l0, l1 = float(g["loss0"]), float(g["loss1"])
    print(f"\n[{tag} fp32 train step] loss0 {losses[0]:.7f} (ref {l0:.7f})  loss1 {losses[1]:.7f} (ref {l1:.7f})  " +
          "  ".join(f"{k} {v:.2e}" if not isinstance(v, str) else f"{k} {v}" for k, v in w.items()) + f"  ({len(names)} gradients)")
    assert abs(losses[0] - l0) < 1e-5 * l0 and abs(losses[1] - l1) < 1e-4 * l1
    assert w["logits"] < FP32_TOL
    assert w["param_norm"] < 1e-4 and w["param_slice"] < 5e-4 and w["bn_buffers"] < 5e-4, w
    assert w["cos"] > 0.9999 and abs(w["norm_ratio"] - 1) < 2e-3, w
    assert w["grad_median"] < 1e-2, w
    if with_oracle32:
        # torch's own float32 run is off by the same order (and differs between hosts: median 2.4e-3 / worst 5.4e-2 in the build
        # container, 4.5e-4 / 1.9e-2 on the GPU box's CPU): a sanity bound, not a pin
        assert w["torch_fp32_median"] > 1e-5 or w["grad_median"] < 1e-3, w
        assert w["grad_max"] < 0.3, w
    return len(names)


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_model_fp32_train_step_vs_reference(mod):
    _train_step_fp32(f"model_tiny_{mod}", mod, "vit_tiny_test", with_oracle32=True)


def test_tiny_model_fp32_train_step_ignore_index_255():
    """the reference's recorded 2-class step with `ignore_index=255` labels (train_multiscale_main.py:30-43)"""
    _train_step_fp32("model_tiny_ignore255", "sam_lora", "vit_tiny_test", ncls=2, ignore=255)


def test_vitb_fp32_train_step_vs_reference():
    """BASELINE configs[0] / [1] at ViT-B width, B = 2: the reference's recorded step, all 333 gradients."""
    assert _train_step_fp32("model_vitb_sam_lora", "sam_lora", "vit_b") == 333


def test_engine_trains_in_fp32_eager_and_graphed():
    """TrainEngine over an fp32 model (flat buffers, parameter bank, deferred gradient delivery, HIP-graph replay): the first two
    losses are the reference's recorded loss0 / loss1, and the captured / replayed steps continue exactly like the eager ones."""
    from ssrs_amd.trainer import TrainEngine
    from ssrs_amd.unetformer_mmsam import UNetFormer
    tag = "model_tiny_sam_lora"
    g = golden(tag)
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles(tag, 2, 256, 6)]
    runs = {}
    for use_graph in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = UNetFormer(num_classes=6, encoder="vit_tiny_test", args=_args("sam_lora"), precision="fp32")
        m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
        m = m.cuda().train()
        m.decoder.segmentation_head[1].p = 0.0
        eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=use_graph)
        runs[use_graph] = [TrainEngine.read_metrics(eng.step(rgb, dsm, tgt).clone())["loss"] for _ in range(5)]
        if use_graph:
            assert len(eng._graphs) == 1
    l0, l1 = float(g["loss0"]), float(g["loss1"])
    print(f"\n[fp32 engine] eager {runs[False]}  graphed {runs[True]}  reference {l0:.7f} {l1:.7f}")
    assert abs(runs[False][0] - l0) < 1e-5 * l0 and abs(runs[False][1] - l1) < 1e-4 * l1
    assert max(abs(a - b) for a, b in zip(runs[False], runs[True])) < 1e-5
    assert runs[False][4] < runs[False][0]
