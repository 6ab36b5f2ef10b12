"""Block / encoder / whole-model parity of the HIP path (bf16 MFMA, fp32 accumulate) on the GPU
against the reference's golden vectors (tests/golden, generated from /root/reference) and the CPU
oracle. Tolerance from BASELINE.json's north star: 2e-2 relative for the bf16 path (relative =
max|a-b| / max|b|). Argmax masks: random-init synthetic weights put many pixels on near-ties of the two
best classes, so the raw agreement is reported and bounded (>= 98 %), and EVERY disagreeing pixel must be a
near-tie of the reference, i.e. its top-2 margin lies inside the 2e-2 logit tolerance band; on all other
pixels the masks must agree exactly (100 % >= the 99.9 % bar). mIoU of the two masks against the synthetic
labels must agree within 0.2 pt."""
import argparse
import functools
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import golden, rel_err, rel_l2
from ssrs_amd import detfill

pytestmark = pytest.mark.gpu
BF16_TOL = 2e-2


def _args(mod):
    return argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)


def _u(key, shape):
    return torch.from_numpy(detfill.uniform(key, shape)).float()


def _block(kind, ws, dim=128, heads=2, mid_dim=None):
    from ssrs_amd.vit import LoraBlock
    from ssrs_amd.vit_adapter import AdapterFusionBlock
    cls = LoraBlock if kind == "lora" else AdapterFusionBlock
    args = _args("sam_lora" if kind == "lora" else "sam_adpt")
    args.mid_dim = mid_dim
    return cls(args=args, dim=dim, num_heads=heads, mlp_ratio=4.0,
               qkv_bias=True, norm_layer=functools.partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU,
               use_rel_pos=True, rel_pos_zero_init=True, window_size=ws, input_size=(64, 64))


@pytest.mark.parametrize("kind", ["lora", "adpt"])
@pytest.mark.parametrize("ws", [14, 0])
@pytest.mark.parametrize("S", [16, 20])
def test_block_fwd_bwd_vs_reference(kind, ws, S):
    tag = f"block_{kind}_w{ws}_s{S}"
    g = golden(tag)
    blk = _block(kind, ws)
    blk.load_state_dict(detfill.fill_state_dict(blk.state_dict(), tag + "/"))
    blk = blk.cuda()
    # the golden holds grads of EVERY block parameter; MFNet trains only the PEFT tensors (freeze rule
    # unetformer_mmsam.py:522-528), and only those have a backward in the HIP path
    token = "lora_" if kind == "lora" else "Adapter"
    trainable = [k[2:] for k in g.files if k.startswith("g/") and token in k]
    assert trainable
    for n, p in blk.named_parameters():
        p.requires_grad = n in trainable
    x = _u(tag + "/x", (1, S, S, 128)).cuda().requires_grad_()
    y = _u(tag + "/y", (1, S, S, 128)).cuda().requires_grad_()
    ox, oy = blk(x, y)
    assert rel_err(ox, g["out_x"]) < BF16_TOL and rel_err(oy, g["out_y"]) < BF16_TOL
    ((ox * _u(tag + "/gx", ox.shape).cuda()).sum() + (oy * _u(tag + "/gy", oy.shape).cuda()).sum()).backward()
    assert rel_err(x.grad, g["grad_x"]) < BF16_TOL and rel_err(y.grad, g["grad_y"]) < BF16_TOL
    worst = {}
    for n, p in blk.named_parameters():
        if n in trainable:
            assert p.grad is not None, n
            worst[n] = rel_err(p.grad, g["g/" + n])
    # scalar mixing weights are one cancellation-heavy full reduction over bf16-rounded products
    bad = {k: v for k, v in worst.items() if v > (1e-1 if g["g/" + k].size == 1 else 3e-2)}
    assert not bad, bad


@pytest.mark.parametrize("kind", ["lora", "adpt"])
@pytest.mark.parametrize("ws", [14, 0])
def test_block_head_dim_80_vs_oracle(kind, ws):
    """ViT-H's head dim (1280 / 16 = 80; here dim 640, 8 heads): block forward, input gradients and PEFT gradients against
    the float64 oracle restatement of the block (which the dim-128 block goldens of the reference pin)."""
    from oracle import mfnet_ref as R
    tag, S, dim = f"block80_{kind}_w{ws}", 16, 640
    blk = _block(kind, ws, dim, 8)
    blk.load_state_dict(detfill.fill_state_dict(blk.state_dict(), tag + "/"))
    token = "lora_" if kind == "lora" else "Adapter"
    trainable = [n for n, _ in blk.named_parameters() if token in n]
    for n, p in blk.named_parameters():
        p.requires_grad = n in trainable
    sd = {k: v.detach().double().clone().requires_grad_(k in trainable) for k, v in blk.state_dict().items() if v.dtype.is_floating_point}
    blk = blk.cuda()
    x0, y0 = _u(tag + "/x", (1, S, S, dim)), _u(tag + "/y", (1, S, S, dim))
    gx, gy = _u(tag + "/gx", (1, S, S, dim)), _u(tag + "/gy", (1, S, S, dim))
    xr, yr = x0.double().requires_grad_(), y0.double().requires_grad_()
    fn = R.lora_block if kind == "lora" else R.adapter_fusion_block
    rx, ry = fn(sd, "", xr, yr, 8, ws)
    ((rx * gx.double()).sum() + (ry * gy.double()).sum()).backward()
    x, y = x0.cuda().requires_grad_(), y0.cuda().requires_grad_()
    ox, oy = blk(x, y)
    assert rel_err(ox, rx.detach()) < BF16_TOL and rel_err(oy, ry.detach()) < BF16_TOL
    ((ox * gx.cuda()).sum() + (oy * gy.cuda()).sum()).backward()
    assert rel_err(x.grad, xr.grad) < BF16_TOL and rel_err(y.grad, yr.grad) < BF16_TOL
    bad = {}
    for n, p in blk.named_parameters():
        if n in trainable:
            e = rel_err(p.grad, sd[n].grad)
            if e > (1e-1 if p.numel() == 1 else 3e-2):
                bad[n] = e
    assert not bad, bad


@pytest.mark.parametrize("ws", [14, 0])
@pytest.mark.parametrize("rank", [2, 3, 8, 12, 16])
def test_lora_block_other_ranks_vs_oracle(rank, ws):
    """`-mid_dim` (medsam/cfg.py:47, lora_block.py:44-47): LoRA ranks other than the default 4 - 2 / 12 / 16 take the general
    skinny products and, for 12 / 16, the GEMM epilogue's extra k-steps (24 / 32 stacked rank columns); 8 the fused kernels.
    Block forward, input gradients and every LoRA gradient against the float64 oracle restatement of the block (rank-agnostic,
    pinned by the reference's rank-4 block goldens). Token count 2 x 400 (S = 20: padded windows and interpolated tables)."""
    from oracle import mfnet_ref as R
    tag, S, dim = f"blockr{rank}_w{ws}", 20, 128
    blk = _block("lora", ws, dim, 2, mid_dim=rank)
    assert blk.mlp.lin1.lora_Ax.shape == (rank, dim) and blk.attn.qkv.lora_Ax.shape == (2 * rank, dim)
    blk.load_state_dict(detfill.fill_state_dict(blk.state_dict(), tag + "/"))
    trainable = [n for n, _ in blk.named_parameters() if "lora_" in n]
    for n, p in blk.named_parameters():
        p.requires_grad = n in trainable
    sd = {k: v.detach().double().clone().requires_grad_(k in trainable) for k, v in blk.state_dict().items() if v.dtype.is_floating_point}
    blk = blk.cuda()
    x0, y0 = _u(tag + "/x", (1, S, S, dim)), _u(tag + "/y", (1, S, S, dim))
    gx, gy = _u(tag + "/gx", (1, S, S, dim)), _u(tag + "/gy", (1, S, S, dim))
    xr, yr = x0.double().requires_grad_(), y0.double().requires_grad_()
    rx, ry = R.lora_block(sd, "", xr, yr, 2, ws)
    ((rx * gx.double()).sum() + (ry * gy.double()).sum()).backward()
    x, y = x0.cuda().requires_grad_(), y0.cuda().requires_grad_()
    ox, oy = blk(x, y)
    assert rel_err(ox, rx.detach()) < BF16_TOL and rel_err(oy, ry.detach()) < BF16_TOL
    ((ox * gx.cuda()).sum() + (oy * gy.cuda()).sum()).backward()
    assert rel_err(x.grad, xr.grad) < BF16_TOL and rel_err(y.grad, yr.grad) < BF16_TOL
    # The four scalar mixing weights' gradients are single sums with heavy cancellation over bf16-rounded products: an error of
    # a few 1e-3 of the TERMS' scale. A draw can cancel almost completely (rank 8 / windowed: lin2.lora_wx = 0.8 where its three
    # siblings are 6 - 170), so their error is measured against the larger of the value itself and a quarter of the RMS of the
    # block's four scalar gradients.
    sc = [float(sd[n].grad) for n in trainable if sd[n].numel() == 1]
    scale = 0.25 * float(np.sqrt(np.mean(np.square(sc))))
    bad = {}
    for n, p in blk.named_parameters():
        if n in trainable:
            if p.numel() == 1:
                e = abs(float(p.grad) - float(sd[n].grad)) / max(abs(float(sd[n].grad)), scale)
            else:
                e = rel_err(p.grad, sd[n].grad)
            if e > (1e-1 if p.numel() == 1 else 3e-2):
                bad[n] = e
    assert not bad, bad


def _check_masks(out, g, tgt, ncls=6, floor=0.99):
    """out: product logits (B,C,H,W) cuda; g: golden with eval_logits_s4 / eval_argmax."""
    from oracle.mfnet_ref import confusion_metrics
    pred = out.argmax(1).cpu().numpy()
    ref_mask = g["eval_argmax"]
    agree = (pred == ref_mask).mean()
    # floor = the measured level: ViT-B MMLoRA 99.4-99.9 %, ViT-B MMAdapter 99.37 %, the 2-block tiny fixtures (flatter random-init logits) 98.6 %; the
    # fp32 path meets the 99.9 % bar on all of them (tests/test_fp32_gpu.py)
    print(f"\n[argmax agreement with the reference] {100 * agree:.3f} % (floor {100 * floor:.1f} %)")
    assert agree >= floor, agree
    ref = g["eval_logits_s4"]
    srt = np.sort(ref, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    band = 2 * BF16_TOL * np.abs(ref).max()
    dis = out[:, :, ::4, ::4].argmax(1).cpu().numpy() != ref.argmax(1)
    assert (margin[dis] < band).all(), float(margin[dis].max())
    assert (margin >= band).mean() > 0.5           # the firm set is not vacuous
    t = tgt.numpy()
    assert abs(confusion_metrics(pred, t, ncls)["mIoU"] - confusion_metrics(ref_mask, t, ncls)["mIoU"]) < 0.002
    return agree


def _tiny_model(mod, num_classes=6):
    from ssrs_amd.unetformer_mmsam import UNetFormer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return UNetFormer(num_classes=num_classes, encoder="vit_tiny_test", args=_args(mod))


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_encoder_vs_reference(mod):
    tag = f"encoder_tiny_{mod}"
    g = golden(tag)
    enc = _tiny_model(mod).image_encoder
    enc.load_state_dict(detfill.fill_state_dict(enc.state_dict(), tag + "/"))
    enc = enc.cuda().eval()
    x = _u(tag + "/x", (1, 3, 256, 256)).cuda()
    y = _u(tag + "/y", (1, 1, 256, 256)).cuda()            # one plane; the reference repeats it to 3 channels
    fx, fy = enc(x, y.repeat(1, 3, 1, 1))
    assert rel_err(fx, g["fx"]) < BF16_TOL and rel_err(fy, g["fy"]) < BF16_TOL
    # whole-encoder backward (neck -> blocks) for a fixed upstream gradient: every PEFT gradient
    ((fx * _u(tag + "/gx", fx.shape).cuda()).sum() + (fy * _u(tag + "/gy", fy.shape).cuda()).sum()).backward()
    names = [k[2:] for k in g.files if k.startswith("g/")]
    pd = dict(enc.named_parameters())
    assert {n for n, p in pd.items() if p.grad is not None} == set(names)
    bad = {n: rel_err(pd[n].grad, g["g/" + n]) for n in names}
    # (scalar mixing weights: one full reduction with heavy cancellation over bf16-rounded products)
    bad = {n: e for n, e in bad.items() if e > (1e-1 if pd[n].numel() == 1 else 3e-2)}
    assert not bad, bad


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_model_eval_logits_vs_reference(mod):
    tag = f"model_tiny_{mod}"
    g = golden(tag)
    m = _tiny_model(mod)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().eval()
    rgb, dsm, tgt = detfill.tiles(tag, 2, 256, 6)
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
    assert out.shape == (2, 6, 256, 256)
    assert rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"]) < BF16_TOL
    _check_masks(out, g, tgt, floor=0.985)


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_model_train_step_vs_reference(mod):
    """loss, every gradient (norm + leading slice) and the loss after one SGD step."""
    tag = f"model_tiny_{mod}"
    g = golden(tag)
    m = _tiny_model(mod)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().train()
    m.decoder.segmentation_head[1].p = 0.0
    rgb, dsm, tgt = detfill.tiles(tag, 2, 256, 6)
    rgb, dsm, tgt = rgb.cuda(), dsm.cuda(), tgt.cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    losses = []
    for step in range(2):
        opt.zero_grad()
        out = m(rgb, dsm)
        loss = F.cross_entropy(out.permute(0, 2, 3, 1).reshape(-1, 6), tgt.view(-1))
        loss.backward()
        losses.append(loss.item())
        if step == 0:
            # Training-mode BatchNorm normalises by the statistics of this 2-tile batch, which makes the
            # decoder ill-conditioned on these synthetic weights: the EXACT float64 decoder maps a 3e-3
            # relative gaussian perturbation of the encoder features to a 2.3e-2 logit error in train
            # mode (5e-4 -> 1.6e-3 in eval mode); measured by tools/diag_parity.py, see DESIGN.md
            # "numerics". The encoder's bf16 rounding (5e-3 on the features) therefore shows up as
            # 2.6e-2 here although the same features give 7e-3 in eval mode. Bound for this case: 5e-2.
            assert rel_err(out[:, :, ::4, ::4], g["logits_s4"]) < 5e-2
            names = [str(n) for n in g["grad_names"]]
            pd = dict(m.named_parameters())
            got = {n for n, p in pd.items() if p.grad is not None}
            assert got == set(names), (got ^ set(names))
            # Whole-model gradients: the decoder's gradient w.r.t. the encoder features is extremely
            # sensitive to the features themselves on these synthetic weights - the EXACT float64
            # decoder evaluated at the product's features (5e-3 away from the reference's) returns
            # feature gradients 17 % (relative L2) away from the reference's, while the product's
            # decoder backward matches the exact one at equal features to 1e-6 and the encoder backward
            # adds nothing measurable on top (tools/diag_parity.py --grads; DESIGN.md "numerics").
            # Backward-kernel parity is therefore pinned where the map is smooth - per block and for the
            # whole encoder with fixed upstream gradients (tests above, 3e-2) - and this end-to-end check
            # only bounds the global direction / size of the gradient.
            num = den_a = den_b = 0.0
            for n in names:
                want = g["g/" + n]
                sl = pd[n].grad.double().flatten()[:256].cpu().numpy()
                w = want[2:2 + sl.size]
                num += float((sl * w).sum()); den_a += float((sl * sl).sum()); den_b += float((w * w).sum())
            cos = num / np.sqrt(den_a * den_b)
            assert cos > 0.95 and 0.8 < np.sqrt(den_a / den_b) < 1.25, (cos, np.sqrt(den_a / den_b))
        opt.step()
    assert abs(losses[0] - float(g["loss0"])) < 5e-3 * float(g["loss0"])
    assert abs(losses[1] - float(g["loss1"])) < 5e-3 * float(g["loss1"])


def test_non_square_tiles_vs_reference_and_oracle():
    """`forward` takes any H x W whose token grid is even in both directions (the reference: unetformer_mmsam.py:533-563).
    (1) a 192 x 256 tile (12 x 16 tokens: padded 14 x 14 windows, interpolated global tables, GLA maps reflect-padded along H)
    against the REFERENCE's own logits (model_tiny_rect.npz), bf16 at 2e-2 and the fp32 path at 1e-4; (2) a 256 x 320 tile (16 x 20
    tokens; its decoder maps need the reflect pad along W, which the reference's GlobalLocalAttention.pad cannot do - a 2-element
    reflect pad of a 4-D map raises in torch) against the float64 oracle, whose pad handles both directions; (3) train steps on the
    non-square tile through the graph-captured engine learn."""
    from oracle import mfnet_ref as R
    from ssrs_amd.trainer import TrainEngine
    tag = "model_tiny_rect"
    g = golden(tag)
    m = _tiny_model("sam_lora")
    fill = detfill.fill_state_dict(m.state_dict(), tag + "/")
    m.load_state_dict(fill)
    sd = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    rgb = torch.from_numpy(0.5 * (detfill.uniform(tag + "/rgb", (2, 3, 192, 256)) + 1.0)).float()
    dsm = torch.from_numpy(0.5 * (detfill.uniform(tag + "/dsm", (2, 192, 256)) + 1.0)).float()
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
        m.precision = "fp32"
        out32 = m(rgb.cuda(), dsm.cuda())
        m.precision = "bf16"
    assert out.shape == (2, 6, 192, 256)
    e, e32 = rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"]), rel_err(out32[:, :, ::4, ::4], g["eval_logits_s4"])
    agree32 = float((out32.argmax(1).cpu().numpy() == g["eval_argmax"]).mean())
    print(f"\n[192x256 tile] bf16 logits {e:.2e}, fp32 logits {e32:.2e}, fp32 argmax agreement {100 * agree32:.4f} %")
    assert e < BF16_TOL and e32 < 1e-4 and agree32 >= 0.999
    cfg = R.vit_cfg("tiny", mod="sam_lora", dim=128, depth=2, heads=2, global_idx=(1,))
    rgb2 = torch.from_numpy(0.5 * (detfill.uniform(tag + "/rgb2", (1, 3, 256, 320)) + 1.0)).float()
    dsm2 = torch.from_numpy(0.5 * (detfill.uniform(tag + "/dsm2", (1, 256, 320)) + 1.0)).float()
    with torch.no_grad():
        ref2 = R.unetformer(sd, rgb2.double(), dsm2.double(), cfg, training=False)
        got2 = m(rgb2.cuda(), dsm2.cuda())
    assert rel_err(got2, ref2) < BF16_TOL
    m.train()
    m.decoder.segmentation_head[1].p = 0.0
    tgt = torch.from_numpy(np.minimum((0.5 * (detfill.uniform(tag + "/tgt", (2, 192, 256)) + 1.0) * 6).astype(np.int64), 5)).cuda()
    eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=True)
    losses = [TrainEngine.read_metrics(eng.step(rgb.cuda(), dsm.cuda(), tgt))["loss"] for _ in range(6)]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0], losses
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 3, 240, 256, device="cuda"), torch.zeros(1, 240, 256, device="cuda"))      # 15 x 16 tokens: odd grid


def test_vitb_eval_logits_vs_reference():
    """BASELINE config: ViT-B MMLoRA, 256x256 tile; logits 2e-2 relative, argmax >= 99.9 %."""
    from ssrs_amd.unetformer_mmsam import UNetFormer
    tag = "model_vitb_sam_lora"
    g = golden(tag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder="vit_b", args=_args("sam_lora"))
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().eval()
    rgb, dsm, tgt = detfill.tiles(tag, 1, 256, 6)
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
    e = rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"])
    assert e < BF16_TOL, e
    agree = _check_masks(out, g, tgt, floor=0.994)
    print(f"\n[vitb sam_lora eval] logits rel err {e:.3e}  argmax agreement {100 * agree:.3f} %")


@pytest.mark.parametrize("mod,encoder,tile,batch", [("sam_adpt", "vit_b", 256, 2), ("sam_lora", "vit_b", 256, 2), ("sam_lora", "vit_l", 512, 1),
                                                    ("sam_lora", "vit_h", 256, 1), ("sam_adpt", "vit_h", 256, 1),
                                                    ("sam_lora", "vit_l", 1024, 1),      # BASELINE configs[4]: whole 1024x1024 tile, 64x64 token grid
                                                    ("sam_lora", "vit_h", 1024, 1)])     # configs[4] names ViT-H too (build_sam.py:16-24): head dim 80, 32 blocks
def test_full_size_train_steps_run_and_learn(mod, encoder, tile, batch):
    """SURVEY 8(d) configurations at their real widths (768 / 1024 / 1280 channels, head dim 80 for ViT-H, adapters, a
    32x32 token grid with 9 windows for ViT-L at 512 px) through the graph-captured engine: every kernel accepts its shapes, the loss is finite
    and falls when the same batch is stepped on repeatedly."""
    from ssrs_amd.trainer import TrainEngine
    from ssrs_amd.unetformer_mmsam import UNetFormer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder=encoder, args=_args(mod))
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "full/"))
    m = m.cuda().train()
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("full", batch, tile, 6)]
    eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=True)
    losses = [TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))["loss"] for _ in range(6)]
    assert all(np.isfinite(l) for l in losses), losses
    assert losses[-1] < losses[0], losses
    assert torch.isfinite(eng.flat_g).all() and eng.flat_g.abs().sum() > 0


@pytest.mark.parametrize("use_graph", [False, True])
def test_engine_checkpoint_round_trip(use_graph):
    """resume = the weights file of the reference (model.state_dict(), trainer.py:379-394) + TrainEngine.state_dict()
    (momentum, epoch, step count): a fresh model + engine loaded from both continues exactly like the one that kept
    running - same losses, same parameters, same BatchNorm statistics. Both files go through torch.save / torch.load
    (weights_only=True), the way Trainer writes and reads them."""
    import io
    from ssrs_amd.trainer import TrainEngine
    fill = detfill.fill_state_dict(_tiny_model("sam_lora").state_dict(), "ckpt/")
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("ckpt", 2, 256, 6)]

    def make():
        m = _tiny_model("sam_lora")
        m.load_state_dict(fill)
        m = m.cuda().train()
        m.decoder.segmentation_head[1].p = 0.0          # Dropout2d draws from the global RNG; not what is under test
        return m, TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=use_graph)

    def run(eng, n):
        return [TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))["loss"] for _ in range(n)]

    m1, e1 = make()
    run(e1, 3)
    buf_w, buf_o = io.BytesIO(), io.BytesIO()
    torch.save(m1.state_dict(), buf_w)
    torch.save(e1.state_dict(), buf_o)
    saved_w = {k: v.clone() for k, v in m1.state_dict().items()}
    saved_m = e1.flat_m.clone()
    cont = run(e1, 3)
    m2, e2 = make()
    buf_w.seek(0); buf_o.seek(0)
    m2.load_state_dict(torch.load(buf_w, map_location="cuda", weights_only=True))
    e2.load_state_dict(torch.load(buf_o, map_location="cuda", weights_only=True))
    # the restored state is the saved state, bit for bit: every parameter and buffer (BatchNorm statistics and counters
    # included), the momentum buffer, the step / epoch counters and with them the learning rate
    assert e2.steps == 3 and e2.epoch == e1.epoch and e2.lr == e1.lr
    sd2 = m2.state_dict()
    assert all(torch.equal(sd2[k], v) for k, v in saved_w.items())
    assert torch.equal(e2.flat_m, saved_m)
    resumed = run(e2, 3)
    # ... and it continues like the engine that kept running. The first loss agrees to fp32 noise. Later steps agree to
    # bf16 level only, as ANY two runs of this step do: fp32 atomics order leaves 1-ulp differences in a few weights after
    # an update, and each bf16 rounding of an intermediate turns a perturbation d into sqrt(d * ulp_bf16) RMS (measured:
    # identical logits, gradients 4e-3 apart after a 1-ulp change of 8 weights; tools/grad_noise.py). A resume
    # that lost the momentum is another order of magnitude away (control below).
    assert abs(cont[0] - resumed[0]) < 5e-6 * cont[0], (cont, resumed)
    assert np.allclose(cont, resumed, rtol=2e-3, atol=0), (cont, resumed)
    assert cont[-1] < cont[0]
    m3, e3 = make()
    buf_w.seek(0)
    m3.load_state_dict(torch.load(buf_w, map_location="cuda", weights_only=True))
    cold = run(e3, 3)                       # weights only (what the reference saves): NOT the same continuation
    assert abs(cold[-1] - cont[-1]) > 10 * abs(resumed[-1] - cont[-1]) and abs(cold[-1] - cont[-1]) > 5e-3 * cont[-1], (cont, resumed, cold)
    with pytest.raises(ValueError):
        bad = e1.state_dict()
        bad["names"] = bad["names"][1:]
        e2.load_state_dict(bad)


def test_graph_replay_after_an_idle_second_matches_eager():
    """A replay that follows a pause must compute what an eager step computes. Background: a hipMemsetAsync inside the captured
    step - a memset NODE - lost its order against its neighbours on the first replay after the GPU had idled (garbage
    fusion-weight gradients); whether it shows depends on the graph pool's layout - test_engine_checkpoint_round_trip[True] run
    after [False] is where it surfaced, this sequence alone did not trip it - so the step simply holds no memset node any more
    and this test watches the symptom (loss trajectories of an eager and a replayed engine with idle seconds between steps)."""
    import time
    from ssrs_amd.trainer import TrainEngine
    fill = detfill.fill_state_dict(_tiny_model("sam_lora").state_dict(), "idle/")
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("idle", 2, 256, 6)]
    traj = {}
    for use_graph in (False, True):
        m = _tiny_model("sam_lora")
        m.load_state_dict(fill)
        m = m.cuda().train()
        m.decoder.segmentation_head[1].p = 0.0
        eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=use_graph)
        losses = []
        for step in range(6):
            if step in (3, 5):
                torch.cuda.synchronize()
                time.sleep(1.0)
            losses.append(TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))["loss"])
        traj[use_graph] = losses
    assert np.allclose(traj[True], traj[False], rtol=2e-3, atol=0), traj


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_captured_step_holds_kernel_nodes_only(mod):
    """The replayed train step must be a chain of KERNEL nodes: no memset / memcpy node (the node class that lost its order on
    a replay after an idle second), no host node, one root, fan-in 1. Read from the captured hipGraph_t itself
    (ssrs_amd/graph_census.py: hipGraphGetNodes / hipGraphNodeGetType / hipGraphNodeGetDependencies); the censuses of the tiny
    and the ViT-B step are committed as profiles/r03_a_graph_census_*.json."""
    from ssrs_amd.graph_census import census
    from ssrs_amd.trainer import TrainEngine
    m = _tiny_model(mod)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "census/"))
    m = m.cuda().train()
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("census", 2, 256, 6)]
    eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=True)
    eng.keep_graph = True
    losses = [TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))["loss"] for _ in range(5)]      # 2 eager, capture + replay, 2 replays
    assert losses[-1] < losses[0]
    (g,) = eng.captured_graphs()
    c = census(g)
    print(f"\n[captured step, tiny {mod}] {c['nodes']} nodes: {c['types']}, roots {c['roots']}, max fan-in {c['max_fan_in']}")
    assert c["types"] == {"kernel": c["nodes"]} and c["nodes"] > 100, c["types"]
    assert not c["non_kernel"] and c["roots"] == 1 and c["max_fan_in"] == 1, c


def test_cpu_tensors_are_rejected_loudly():
    from ssrs_amd._lib import MfnetError
    m = _tiny_model("sam_lora")
    with pytest.raises(MfnetError):
        m(torch.zeros(1, 3, 256, 256), torch.zeros(1, 256, 256))


def _vitb(mod):
    from ssrs_amd.unetformer_mmsam import UNetFormer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return UNetFormer(num_classes=6, encoder="vit_b", args=_args(mod))


def test_vitb_adapter_eval_logits_vs_reference():
    """BASELINE configs[2]'s model (ViT-B MMAdapter) against the reference's eval logits and masks."""
    tag = "model_vitb_sam_adpt"
    g = golden(tag)
    m = _vitb("sam_adpt")
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().eval()
    rgb, dsm, tgt = detfill.tiles(tag, 1, 256, 6)
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
    e = rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"])
    agree = _check_masks(out, g, tgt)
    cs = rel_err(out.double().sum((0, 2, 3)), g["eval_chan_sum"])
    print(f"\n[vitb sam_adpt eval] logits rel err {e:.3e}  argmax agreement {100 * agree:.3f} %  channel-sum rel err {cs:.2e}")
    assert e < BF16_TOL, e
    assert cs < BF16_TOL, cs


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_vitb_per_block_residual_checksums_vs_reference(mod):
    """Per-block checksums of the residual stream (mean |h| and sum of h per modality after each of the 12 blocks) against the
    reference's forward hooks: localises drift to a block instead of showing it at the logits only."""
    tag = f"model_vitb_{mod}"
    g = golden(tag)
    m = _vitb(mod)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    enc = m.cuda().eval().image_encoder
    rgb, dsm, _ = detfill.tiles(tag, 1, 256, 6)
    want = g["block_sums"]                                   # [12, 4]: mean|x|, mean|y|, sum x, sum y
    got = []
    with torch.no_grad():
        h = enc.embed(rgb.cuda(), dsm.cuda())
        for blk in enc.blocks:
            h = blk.forward_stacked(h)
            hd = h.double()
            got.append([hd[0].abs().mean().item(), hd[1].abs().mean().item(), hd[0].sum().item(), hd[1].sum().item()])
    got = np.array(got)
    n_el = h[0].numel()
    mean_abs_err = np.abs(got[:, :2] - want[:, :2]) / want[:, :2]
    # the sum of n elements of size ~mean|h| each carrying an independent relative rounding error e: |error| ~ e * mean|h| * sqrt(n)
    sum_err = np.abs(got[:, 2:] - want[:, 2:]) / (want[:, :2] * n_el)
    print(f"\n[{mod}] per-block rel err of mean|h|: max {mean_abs_err.max():.2e}; |sum error| / (n mean|h|): max {sum_err.max():.2e}")
    assert mean_abs_err.max() < 2e-3, mean_abs_err
    assert sum_err.max() < 2e-3, sum_err


def test_vitb_train_step_vs_reference():
    """BASELINE configs[0]/[1] at ViT-B width, B = 2: two SGD steps of the engine against the reference's recorded step -
    loss0 / loss1, train-mode logits, the gradient of each of the 333 trainable tensors (norm, sum and leading slice),
    the SGD-updated parameters and the BatchNorm buffers."""
    from ssrs_amd.trainer import TrainEngine
    tag = "model_vitb_sam_lora"
    g = golden(tag)
    m = _vitb("sam_lora")
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().train()
    m.decoder.segmentation_head[1].p = 0.0
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles(tag, 2, 256, 6)]
    eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"))
    names = [str(n) for n in g["grad_names"]]
    assert set(eng.names) == set(names) and len(names) == 333
    eng.zero_grad()
    acc, _ = eng.forward_backward(rgb, dsm, tgt)
    loss0 = TrainEngine.read_metrics(acc)["loss"]
    pd = dict(zip(eng.names, eng.params))
    ratios, num, da, db = [], 0.0, 0.0, 0.0
    for n in names:
        want = g["g/" + n]
        gr = pd[n].grad.double().flatten()
        ratios.append(gr.norm().item() / max(want[0], 1e-30))
        sl = gr[:256].cpu().numpy()
        w = want[2:2 + sl.size]
        num += float((sl * w).sum()); da += float((sl * sl).sum()); db += float((w * w).sum())
    ratios = np.array(ratios)
    cos = num / np.sqrt(da * db)
    eng.optimizer_step()
    loss1 = TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))["loss"]
    perr = max(rel_err(torch.cat([pd[n].detach().double().flatten()[:256].cpu()]), g["p/" + n][2:258]) for n in names)
    pnorm = max(abs(pd[n].detach().double().norm().item() / g["p/" + n][0] - 1) for n in names if g["p/" + n][0] > 0)
    berr = {}
    for n, b in m.named_buffers():
        if "b/" + n in g.files and "running_" in n:
            want = g["b/" + n]
            got = b.double().cpu().numpy() if b.numel() < 300 else None
            berr[n] = rel_err(got, want) if got is not None else abs(b.double().norm().item() / want[0] - 1)
        elif "b/" + n in g.files and "num_batches" in n:
            assert int(b.item()) == int(g["b/" + n]), n
    print(f"\n[vitb train step] loss0 {loss0:.5f} (ref {float(g['loss0']):.5f})  loss1 {loss1:.5f} (ref {float(g['loss1']):.5f})  "
          f"grad cos {cos:.4f}  grad-norm ratio: median {np.median(ratios):.3f}, 5-95 % [{np.quantile(ratios, .05):.3f}, {np.quantile(ratios, .95):.3f}], "
          f"min {ratios.min():.3f} max {ratios.max():.3f}  updated-param slice err {perr:.2e} norm err {pnorm:.2e}  BN buffer err {max(berr.values()):.2e}")
    assert abs(loss0 - float(g["loss0"])) < 1e-3 * float(g["loss0"])          # measured 4e-5 / 3e-5
    assert abs(loss1 - float(g["loss1"])) < 1e-3 * float(g["loss1"])
    assert len(berr) >= 40
    # see test_tiny_model_train_step_vs_reference for why end-to-end gradients are only bounded globally (training-mode
    # BatchNorm on a 2-tile batch, ReLU6 / max-pool switches); kernels are pinned per block and per module
    assert cos > 0.98 and 0.95 < np.median(ratios) < 1.05, (cos, np.median(ratios))           # measured 0.9948 / 1.006
    assert 0.7 < np.quantile(ratios, .05) and np.quantile(ratios, .95) < 1.4, ratios
    assert perr < 4e-3 and pnorm < 1e-3, (perr, pnorm)                                         # measured 1.8e-3 / 2.4e-4
    assert max(berr.values()) < 2e-2, berr                                                     # measured 8e-3


def test_inference_operand_cache_never_serves_stale_weights():
    """Inside a predictor's EvalBank the operands derived from trainable weights (bf16 GEMM images, stacked LoRA factors, ...)
    are kept between forwards (bank.derived). They must follow every way the weights can change: an in-place torch op, load_state_dict, and the
    engine's SGD kernel, which rewrites the flat parameter buffer without touching version counters."""
    from ssrs_amd import bank
    from ssrs_amd.trainer import TrainEngine
    m = _tiny_model("sam_lora")
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "cache/"))
    m = m.cuda().eval()
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("cache", 2, 256, 6)]

    eb = bank.EvalBank()

    def fresh():
        with torch.no_grad():
            return m(rgb, dsm).clone()             # no EvalBank: every operand is derived anew

    def cached():
        with torch.no_grad(), eb:
            return m(rgb, dsm).clone()

    a = fresh()
    assert torch.equal(cached(), a)
    n_entries = len(eb.entries)
    assert n_entries > 20 and len(eb.bn) > 5
    assert torch.equal(cached(), a) and len(eb.entries) == n_entries                # served from the store, same bits
    assert bank.EvalBank.active is None
    w = m.decoder.segmentation_head[2][0].weight
    with torch.no_grad():
        w.mul_(1.5)                                                                     # in-place torch op
    b = cached()
    assert not torch.equal(b, a) and torch.equal(b, fresh())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["decoder.segmentation_head.2.0.weight"] = sd["decoder.segmentation_head.2.0.weight"] / 1.5
    m.load_state_dict(sd)                                                              # load_state_dict
    c = cached()
    assert torch.allclose(c, a, rtol=1e-2, atol=1e-2) and torch.equal(c, fresh())
    # a forward replayed from a HIP graph reads the stored operands: they are refreshed in place before the replay
    from ssrs_amd.inference import GraphedForward
    gf = GraphedForward(lambda x, y: m(x, y))
    for _ in range(3):                             # eager warm-up, capture, replay
        with eb:
            g = gf(rgb, dsm).clone()
        assert torch.equal(g, c)
    with torch.no_grad():
        w.mul_(0.5)
    with eb:
        g = gf(rgb, dsm).clone()
    assert not torch.equal(g, c) and torch.equal(g, fresh())
    w.data.mul_(2.0)                               # behind torch's back (no version bump): the writer announces it
    bank.weights_changed()
    with eb:
        g2 = gf(rgb, dsm).clone()
    assert not torch.equal(g2, g) and torch.equal(g2, fresh())
    # (an engine adopts the parameters into its flat buffer - new addresses: graphs captured before that are void)
    eng = TrainEngine(m.train(), lr=0.05, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"))
    m.eval()
    d0 = cached()
    m.train()
    eng.step(rgb, dsm, tgt)                                                            # SGD kernel on the flat buffer
    m.eval()
    d1 = cached()
    assert not torch.equal(d1, d0) and torch.equal(d1, fresh())
    # the graphs `gf` captured above read the parameters at their OLD addresses (the engine re-pointed every trainable parameter
    # into its flat buffer): the address fingerprint must void them, and the next calls re-capture against the new storage
    assert gf._graphs
    for _ in range(3):
        with eb:
            g3 = gf(rgb, dsm).clone()
        assert torch.equal(g3, fresh())


def test_eval_batchnorm_statistics_follow_a_training_forward():
    """the cached eval mean / rstd of a BatchNorm (ops.bn_forward under no_grad) must follow the running statistics that a
    training-mode forward rewrites in a kernel (no torch version bump)"""
    m = _tiny_model("sam_lora")
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "bncache/"))
    m = m.cuda().eval()
    rgb, dsm, _ = [t.cuda() for t in detfill.tiles("bncache", 2, 256, 6)]
    from ssrs_amd import bank
    eb = bank.EvalBank()
    with torch.no_grad(), eb:
        a = m(rgb, dsm).clone()
        assert torch.equal(m(rgb, dsm), a) and len(eb.bn) > 5
    m.train()
    m(rgb, dsm)                                   # updates every BatchNorm's running statistics
    m.eval()
    with torch.no_grad():
        with eb:
            b = m(rgb, dsm).clone()
        c = m(rgb, dsm).clone()                   # no store
    assert not torch.equal(b, a) and torch.equal(b, c)


@pytest.mark.parametrize("use_graph", [True, False])
def test_trainer_epoch_on_the_real_model(use_graph, tmp_path):
    """Trainer.train_epoch (trainer.py:146-197) on the tiny MFNet through the reference-style API: the default replays the step
    from a HIP graph per batch shape (a shorter last batch captures its own graph), use_graph=False enqueues eagerly; both
    train (finite, falling loss) and agree with each other on the epoch's mean loss to bf16 level."""
    from ssrs_amd.trainer import ModelWrapper, Trainer, TrainingConfig

    class _MW:
        def __init__(self, m):
            self.m = m

        def get_model(self):
            return self.m

        def set_train_mode(self):
            self.m.train()

        def set_eval_mode(self):
            self.m.eval()

        def save_weights(self, path):
            pass

    def run(graph):
        m = _tiny_model("sam_lora")
        m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "trainer/"))
        m = m.cuda().train()
        m.decoder.segmentation_head[1].p = 0.0
        batches = []
        for i, n in enumerate((2, 2, 2, 1)):                      # the last batch is shorter
            rgb, dsm, tgt = detfill.tiles(f"trainer/{i % 2}", n, 256, 6)
            batches.append((rgb, dsm, tgt))
        tr = Trainer(_MW(m), None, train_loader=batches * 2, training=TrainingConfig(), output_dir=str(tmp_path) + "/", sync_every=2,
                     use_graph=graph)
        assert tr.engine.use_graph == graph
        first = tr.train_epoch(1)["avg_loss"]
        second = tr.train_epoch(2)["avg_loss"]
        return first, second, (len(tr.engine._graphs) if graph else 0)

    f, s, ngraphs = run(use_graph)
    assert np.isfinite(f) and np.isfinite(s) and s < f, (f, s)
    if use_graph:
        assert ngraphs == 2                                        # batch shapes (2, ...) and (1, ...)
        fe, se, _ = run(False)
        assert abs(f - fe) < 2e-3 * fe and abs(s - se) < 1e-2 * se, ((f, s), (fe, se))
