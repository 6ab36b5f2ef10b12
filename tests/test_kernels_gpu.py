"""Per-kernel parity on the GPU: HIP kernel (through the C ABI) vs a plain fp32/fp64 PyTorch
restatement of the same op on the SAME bf16-rounded inputs. Tolerances: fp32 accumulate over bf16
inputs -> 2e-3 relative on bf16 outputs (one bf16 rounding), 1e-4..1e-5 on fp32 outputs."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import bf16r, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from ssrs_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1).mul_(scale).cuda()


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,C", [(8192, 768), (1000, 256), (37, 128), (512, 1280), (16384, 256), (4096, 64)])
def test_layernorm_fwd_bwd(ops, rows, C):
    x = rnd(rows, C, seed=1, scale=3.0) + 0.5
    g = 1 + 0.1 * rnd(C, seed=2)
    b = 0.1 * rnd(C, seed=3)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, out_dtype=torch.float32)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-6)
    assert rel_err(y, ref) < 1e-5
    yb, _, _ = ops.layernorm_fwd(x, g, b, 1e-6, out_dtype=torch.bfloat16)
    assert rel_err(yb.float(), ref) < 5e-3
    # backward (fp32 dy and bf16 dy), with residual-gradient add and weight grads
    dy = rnd(rows, C, seed=4)
    dres = rnd(rows, C, seed=5)
    xd = x.double().requires_grad_()
    gd = g.double().requires_grad_()
    bd = b.double().requires_grad_()
    F.layer_norm(xd, (C,), gd, bd, 1e-6).backward(dy.double())
    dg = torch.zeros(C, device="cuda")
    db = torch.zeros(C, device="cuda")
    dx, dxb = ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dres, want_f32=True, want_bf16=True, dgamma=dg, dbeta=db)
    assert rel_err(dx, xd.grad + dres.double()) < 2e-5
    assert rel_err(dxb.float(), xd.grad + dres.double()) < 5e-3
    assert rel_err(dg, gd.grad) < 1e-4 and rel_err(db, bd.grad) < 1e-4
    dyb = dy.to(torch.bfloat16)
    dx2, _ = ops.layernorm_bwd(dyb, x, mean, rstd, g)
    xd.grad = None
    F.layer_norm(xd, (C,), gd, bd, 1e-6).backward(dyb.double())
    assert rel_err(dx2, xd.grad) < 2e-5


@pytest.mark.parametrize("rows,C,R,mix", [(512, 768, 8, False), (512, 768, 4, True), (36, 128, 4, True), (100, 1280, 8, False), (8192, 768, 4, True),
                                          (64, 1024, 4, False)])
def test_layernorm_fwd_with_lora_down_projection(ops, rows, C, R, mix):
    """LayerNorm + a = y @ A_g^T (+ the cross-modal mix) in one launch == the three separate kernels' arithmetic: y and the
    statistics bit-equal to layernorm_fwd, a / U against float64 products of the SAME bf16 y and bf16 A, and close to
    what skinny_nt + lora_mix_fwd give."""
    x = rnd(rows, C, seed=1, scale=3.0) + 0.5
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    A = 0.05 * rnd(2, R, C, seed=4)
    wx, wy = torch.tensor([0.3], device="cuda"), torch.tensor([0.8], device="cuda")
    y0, m0, r0 = ops.layernorm_fwd(x, g, b, 1e-6)
    y, m, r, a, U = ops.layernorm_fwd_lora(x, g, b, 1e-6, A, wx if mix else None, wy if mix else None)
    assert torch.equal(y, y0) and torch.equal(m, m0) and torch.equal(r, r0)
    half = rows // 2
    yd = y.double()
    Ab = A.to(torch.bfloat16).double()                      # the kernel keeps A as bf16 in LDS, like skinny_nt's fragments
    ax, ay = yd[:half] @ Ab[0].t(), yd[half:] @ Ab[1].t()
    assert rel_err(a, torch.cat([ax, ay])) < 1e-5
    if mix:
        want = torch.cat([torch.cat([0.3 * ax, 0.7 * ay], 1), torch.cat([(1 - 0.8) * ax, 0.8 * ay], 1)])
        assert rel_err(U, want) < 1e-5
    else:
        assert U is None
    a_sep = ops.skinny_nt(y, A, rows_per_group=half) if half % 16 == 0 else None
    if a_sep is not None:
        assert rel_err(a, a_sep) < 1e-5
    a1 = ops.layernorm_fwd_lora(x, g, b, 1e-6, A[:1].contiguous())[3]                 # one group: every row uses A[0]
    assert rel_err(a1, yd @ Ab[0].t()) < 1e-5


@pytest.mark.parametrize("rows,C,R,mix", [(8192, 3072, 4, True), (512, 768, 8, False), (36, 128, 4, True), (64, 1024, 8, False),
                                          (8192, 4096, 4, True), (1040, 4096, 4, False)])        # C = 4096: the 64-96 KB LDS image (ViT-L lin2)
def test_lora_down_projection_and_mix_in_one_launch(ops, rows, C, R, mix):
    x = rnd(rows, C, seed=1).to(torch.bfloat16)
    A = 0.05 * rnd(2, R, C, seed=4)
    wx, wy = torch.tensor([0.3], device="cuda"), torch.tensor([0.8], device="cuda")
    a, U = ops.lora_down_mix(x, A, wx if mix else None, wy if mix else None)
    half = rows // 2
    xd, Ab = x.double(), A.to(torch.bfloat16).double()
    ax, ay = xd[:half] @ Ab[0].t(), xd[half:] @ Ab[1].t()
    assert rel_err(a, torch.cat([ax, ay])) < 1e-5
    if mix:
        assert rel_err(U, torch.cat([torch.cat([0.3 * ax, 0.7 * ay], 1), torch.cat([(1 - 0.8) * ax, 0.8 * ay], 1)])) < 1e-5
        if half % 16 == 0:
            assert rel_err(U, ops.lora_mix_fwd(ops.skinny_nt(x, A, rows_per_group=half).contiguous(), wx, wy)) < 1e-5
    else:
        assert U is None


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (1024, 2304, 768), (512, 768, 3072), (800, 192, 768),
                                   (333, 256, 2304), (128, 128, 64), (5000, 3072, 768)])
def test_gemm_plain(ops, M, N, K):
    a = rnd(M, K, seed=1).to(torch.bfloat16)
    w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    ref = a.double() @ w.double().t() + bias.double()
    out = ops.gemm(a, w, bias=bias, out_dtype=torch.float32)
    assert rel_err(out, ref) < 1e-5
    outb = ops.gemm(a, w, bias=bias, out_dtype=torch.bfloat16)
    assert rel_err(outb.float(), ref) < 5e-3


def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric W catches a transposed / permuted C-write (guide section 3)."""
    K = N = 256
    a = torch.eye(K, device="cuda").to(torch.bfloat16)
    w = (torch.arange(N * K, device="cuda").reshape(N, K) % 251).float().to(torch.bfloat16)
    out = ops.gemm(a, w, out_dtype=torch.float32)
    assert torch.equal(out, w.float().t().contiguous())


@pytest.mark.parametrize("R,G", [(8, 1), (8, 2), (4, 2), (16, 1), (24, 2), (32, 2)])
def test_gemm_lowrank_gelu_residual(ops, R, G):
    M, N, K = 1024, 768, 768
    a = rnd(M, K, seed=1).to(torch.bfloat16)
    w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    U = rnd(M, R, seed=4)
    V = rnd(G, N, R, seed=5)
    res = rnd(M, N, seed=6)
    rpg = M // G
    lr = 0.25
    low = torch.cat([U[g * rpg:(g + 1) * rpg].double() @ V[g].double().t() for g in range(G)]) * lr
    pre_ref = a.double() @ w.double().t() + bias.double() + low
    # act=1 (GELU) with pre-activation output, bf16 out
    out, pre = ops.gemm(a, w, bias=bias, U=U, V=V, rows_per_group=rpg, n_groups=G, lr_scale=lr, act=1, want_pre=True)
    assert rel_err(pre.float(), pre_ref) < 5e-3
    assert rel_err(out.float(), F.gelu(pre_ref)) < 5e-3
    # residual add into fp32 (in place: residual aliases out)
    out2 = res.clone()
    ops.gemm(a, w, bias=bias, U=U, V=V, rows_per_group=rpg, n_groups=G, lr_scale=lr, residual=out2, out=out2)
    assert rel_err(out2, pre_ref + res.double()) < 1e-5
    # act=2: multiply by gelu'(aux)
    aux = rnd(M, N, seed=7, scale=2.0).to(torch.bfloat16)
    ad = aux.double().requires_grad_()
    F.gelu(ad).sum().backward()
    out3 = ops.gemm(a, w, act=2, aux=aux, out_dtype=torch.float32)
    assert rel_err(out3, (a.double() @ w.double().t()) * ad.grad) < 2e-5
    # broadcast residual rows (pos-embed add): residual row = m % mod
    pos = rnd(256, N, seed=8)
    out4 = ops.gemm(a, w, residual=pos, res_row_mod=256, out_dtype=torch.float32)
    assert rel_err(out4, a.double() @ w.double().t() + pos.double().repeat(M // 256, 1)) < 1e-5


@pytest.mark.parametrize("tile,N", [(3, 192), (4, 256), (6, 384), (8, 512), (9, 576), (-1, 256)])
@pytest.mark.parametrize("K", [64, 448])
def test_gemm_tile_shapes_all_epilogues(ops, tile, N, K):
    """Every tile shape of the 256-row kernel (tile = column fragments per wave: 256 x 32*tile tiles, lanes own runs of 16 / 8 / 4
    columns) and the 128x128 kernel (-1) through every epilogue, against fp64: bias + two-group rank-8 update + GELU with the
    pre-activation, x GELU'(aux), fp32 residual in place, ReLU6 (3), ReLU6 mask of aux (4), ReLU6 after the residual (5), a
    rank-16 and a rank-4 update, broadcast residual rows; and A = I with an asymmetric W (exact)."""
    M = 512
    a = rnd(M, K, seed=1).to(torch.bfloat16)
    w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    U, V = rnd(M, 8, seed=4), rnd(2, N, 8, seed=5)
    res = rnd(M, N, seed=6)
    aux = rnd(M, N, seed=7, scale=3.0).to(torch.bfloat16)
    ad, wd = a.double(), w.double()
    base = ad @ wd.t()
    low = torch.cat([U[g * 256:(g + 1) * 256].double() @ V[g].double().t() for g in range(2)]) * 0.25
    lora = dict(U=U, V=V, rows_per_group=256, n_groups=2, lr_scale=0.25)
    out, pre = ops.gemm(a, w, bias=bias, act=1, want_pre=True, tile=tile, **lora)
    ref = base + bias.double() + low
    assert rel_err(pre.float(), ref) < 5e-3 and rel_err(out.float(), F.gelu(ref)) < 5e-3
    x = aux.double().requires_grad_()
    F.gelu(x).sum().backward()
    out = ops.gemm(a, w, act=2, aux=aux, out_dtype=torch.float32, tile=tile, **lora)
    assert rel_err(out, (base + low) * x.grad) < 2e-5
    out = res.clone()
    ops.gemm(a, w, bias=bias, residual=out, out=out, tile=tile, **lora)
    assert rel_err(out, ref + res.double()) < 1e-5
    out = ops.gemm(a, w, bias=bias, act=3, tile=tile)
    assert rel_err(out.float(), (base + bias.double()).clamp(0, 6)) < 5e-3
    mask = ((aux.double() > 0) & (aux.double() < 6)).double()
    out = ops.gemm(a, w, act=4, aux=aux, out_dtype=torch.float32, tile=tile)
    assert rel_err(out, base * mask) < 1e-5
    out = ops.gemm(a, w, act=5, residual=res, out_dtype=torch.float32, tile=tile)
    assert rel_err(out, (base + res.double()).clamp(0, 6)) < 1e-5
    U16, V16 = rnd(M, 16, seed=8), rnd(N, 16, seed=9)
    out = ops.gemm(a, w, U=U16, V=V16, lr_scale=0.5, out_dtype=torch.float32, tile=tile)
    assert rel_err(out, base + 0.5 * U16.double() @ V16.double().t()) < 2e-5
    U4, V4 = rnd(M, 4, seed=10), rnd(N, 4, seed=11)
    out = ops.gemm(a, w, U=U4, V=V4, out_dtype=torch.float32, tile=tile)
    assert rel_err(out, base + U4.double() @ V4.double().t()) < 2e-5
    for Rw in (12, 24, 32):              # LoRA ranks 12 / 16: up to 32 stacked rank columns, 8 per extra k-step of the epilogue
        Uw, Vw = rnd(M, Rw, seed=20 + Rw), rnd(2, N, Rw, seed=21 + Rw)
        loww = torch.cat([Uw[g * 256:(g + 1) * 256].double() @ Vw[g].double().t() for g in range(2)]) * 0.5
        out = ops.gemm(a, w, U=Uw, V=Vw, rows_per_group=256, n_groups=2, lr_scale=0.5, out_dtype=torch.float32, tile=tile)
        assert rel_err(out, base + loww) < 2e-5, Rw
    pos = rnd(256, N, seed=12)
    out = ops.gemm(a, w, residual=pos, res_row_mod=256, out_dtype=torch.float32, tile=tile)
    assert rel_err(out, base + pos.double().repeat(2, 1)) < 1e-5
    Ki = (N + 63) // 64 * 64
    eye = torch.zeros((Ki + 255) // 256 * 256, Ki, device="cuda")
    eye[:N, :N] = torch.eye(N, device="cuda")
    wi = torch.zeros(N, Ki, device="cuda")
    wi[:, :N] = (torch.arange(N * N, device="cuda").reshape(N, N) % 251).float()
    out = ops.gemm(eye.to(torch.bfloat16), wi.to(torch.bfloat16), out_dtype=torch.float32, tile=tile)
    assert torch.equal(out[:N], wi[:, :N].t().contiguous())


def test_gemm_encoder_shapes_take_the_big_tiles_and_match(ops):
    """The shapes of the ViT-B step at 16 tiles (M = 8192): the library's own choice (256-row tiles) against the 128x128 kernel."""
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304)):
        a = rnd(8192, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        bias = rnd(N, seed=3)
        got = ops.gemm(a, w, bias=bias, out_dtype=torch.float32)
        ref = ops.gemm(a, w, bias=bias, out_dtype=torch.float32, tile=-1)
        assert rel_err(got, ref) < 1e-5, (N, K)
    from ssrs_amd._lib import MfnetError
    with pytest.raises(MfnetError):
        ops.gemm(rnd(100, 64).to(torch.bfloat16), rnd(96, 64).to(torch.bfloat16), tile=3)          # M % 256 != 0


def test_gemm_rejects_bad_shapes(ops):
    from ssrs_amd._lib import MfnetError
    a = torch.zeros(64, 100, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(64, 100, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(MfnetError):
        ops.gemm(a, w)
    with pytest.raises(MfnetError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(16, 64, dtype=torch.bfloat16))


# ------------------------------------------------------------------------------------------ skinny / lora
@pytest.mark.parametrize("M,K,R", [(8192, 768, 8), (4100, 3072, 4), (100, 128, 16), (1024, 768, 24), (800, 384, 32), (512, 128, 12)])
def test_skinny(ops, M, K, R):          # R = 12 / 24 / 32: the column-chunked forms (LoRA ranks 12 / 16)
    x = rnd(M, K, seed=1).to(torch.bfloat16)
    wt = rnd(R, K, seed=2)
    out = ops.skinny_nt(x, wt, scale=0.5)
    # the rank-R matrix is rounded to bf16 fragments for the MFMA: exact against the rounded operand
    assert rel_err(out, 0.5 * (x.double() @ bf16r(wt).double().t())) < 1e-5
    assert rel_err(out, 0.5 * (x.double() @ wt.double().t())) < 5e-3
    if M % 32 == 0:     # per-modality matrices: rows [0, M/2) use wt2[0], the rest wt2[1]
        wt2 = torch.stack([wt, rnd(R, K, seed=9)])
        o2 = ops.skinny_nt(x, wt2, rows_per_group=M // 2)
        ref2 = torch.cat([x[:M // 2].double() @ bf16r(wt2[0]).double().t(), x[M // 2:].double() @ bf16r(wt2[1]).double().t()])
        assert rel_err(o2, ref2) < 1e-5
    u = rnd(M, R, seed=3)
    for tr in (False, True):
        o = ops.skinny_tn(x, u, transposed=tr, scale=2.0)
        ref = 2.0 * (x.double().t() @ u.double())
        assert rel_err(o, ref.t() if tr else ref) < 1e-4
        if M % 2 == 0:
            o2 = ops.skinny_tn(x, u, transposed=tr, scale=2.0, n_groups=2)
            for g in range(2):
                sl = slice(g * M // 2, (g + 1) * M // 2)
                rg = 2.0 * (x[sl].double().t() @ u[sl].double())
                assert rel_err(o2[g], rg.t() if tr else rg) < 1e-4


@pytest.mark.parametrize("M,C,G,R,RP", [(8192, 768, 1, 8, 8), (8192, 2304, 2, 8, 8), (4096, 3072, 2, 4, 0), (1024, 1280, 2, 8, 8), (200, 64, 1, 8, 4),
                                        (96, 320, 2, 4, 8), (8192, 3072, 1, 0, 8)])
def test_lora_side_fused_products(ops, M, C, G, R, RP):
    """mfnet_lora_side: both rank-r products of a LoRA backward from one pass over the gradient matrix - the row products
    P = y . Vt[g]^T as partial sums over 256-column strips, the column products dV = y^T . u as partial tiles over row splits
    (summed by the gradient delivery) - per modality group, ragged row counts and a last strip narrower than 256 included."""
    y = rnd(M, C, seed=1).to(torch.bfloat16)
    U = rnd(M, R, seed=2) if R else None
    Vt = (rnd(G, RP, C, seed=3) if G > 1 else rnd(RP, C, seed=3)) if RP else None
    dv, pp = ops.lora_side(y, U=U, Vt=Vt, n_groups=G, scale=0.5)
    Mg = M // G
    if RP:
        assert tuple(pp.shape) == ((C + 255) // 256, M, 8)
        P = ops.reduce_parts(pp, 2.0)
        vt = Vt if G > 1 else Vt[None]
        ref = torch.cat([y[g * Mg:(g + 1) * Mg].double() @ bf16r(vt[g]).double().t() for g in range(G)]) * 2.0
        assert rel_err(P[:, :RP], ref) < 1e-5
        assert float(P[:, RP:].abs().max()) == 0.0 if RP < 8 else True
    else:
        assert pp is None
    if R:
        got = dv.materialize()
        got = got if G > 1 else got[None]
        for g in range(G):
            sl = slice(g * Mg, (g + 1) * Mg)
            assert rel_err(got[g], 0.5 * (y[sl].double().t() @ U[sl].double())) < 1e-4, g
        if G > 1:       # the views the block backward hands to the gradient delivery
            assert rel_err(dv[1].t().materialize(), 0.5 * (y[Mg:].double().t() @ U[Mg:].double()).t()) < 1e-4
    else:
        assert dv is None
    # partial sums feed lora_mix_bwd directly
    if RP == 8 and G == 1 and R == 8 and M % 2 == 0:
        a = rnd(M, 4, seed=5)
        wx, wy = torch.tensor([0.3], device="cuda"), torch.tensor([0.9], device="cuda")
        da1, dw1 = ops.lora_mix_bwd(pp, a, wx, wy)
        da2, dw2 = ops.lora_mix_bwd(ops.reduce_parts(pp), a, wx, wy)
        assert rel_err(da1, da2) < 1e-6 and rel_err(dw1, dw2) < 1e-4


def test_lora_mix(ops):
    M, r = 2048, 4
    a = rnd(M, r, seed=1)
    wx = torch.tensor([0.37], device="cuda")
    wy = torch.tensor([0.81], device="cuda")
    U = ops.lora_mix_fwd(a, wx, wy)
    ax, ay = a[:M // 2], a[M // 2:]
    ref = torch.cat([torch.cat([wx * ax, (1 - wx) * ay], 1), torch.cat([(1 - wy) * ax, wy * ay], 1)])
    assert rel_err(U, ref) < 1e-6
    P = rnd(M, 2 * r, seed=2)
    ad = a.double().requires_grad_()
    wxd = wx.double().requires_grad_()
    wyd = wy.double().requires_grad_()
    axd, ayd = ad[:M // 2], ad[M // 2:]
    Ud = torch.cat([torch.cat([wxd * axd, (1 - wxd) * ayd], 1), torch.cat([(1 - wyd) * axd, wyd * ayd], 1)])
    (Ud * P.double()).sum().backward()
    da, dw = ops.lora_mix_bwd(P, a, wx, wy)
    assert rel_err(da, ad.grad) < 1e-5
    assert rel_err(dw, torch.stack([wxd.grad[0], wyd.grad[0]])) < 1e-4


# ------------------------------------------------------------------------------------------ im2col
def test_im2col_patch_and_3x3(ops):
    B, H, W = 2, 64, 96
    rgb = rnd(B, 3, H, W, seed=1)
    dsm = rnd(B, H, W, seed=2)
    cols = ops.im2col_patch(rgb, dsm, 16)
    both = torch.cat([rgb, dsm[:, None].repeat(1, 3, 1, 1)])
    ref = F.unfold(both, 16, stride=16).transpose(1, 2).reshape(-1, 768)
    assert torch.equal(cols.float(), bf16r(ref))
    y3 = rnd(B, 3, H, W, seed=5)                                    # a real 3-channel second modality
    ref = F.unfold(torch.cat([rgb, y3]), 16, stride=16).transpose(1, 2).reshape(-1, 768)
    assert torch.equal(ops.im2col_patch(rgb, y3, 16).float(), bf16r(ref))
    with pytest.raises(ValueError):
        ops.im2col_patch(rgb, rnd(B, 2, H, W, seed=6), 16)
    G, S, C = 3, 10, 64
    x = rnd(G, S, S, C, seed=3).to(torch.bfloat16)
    c3 = ops.im2col3x3(x, G, S, S)
    xr = x.float().permute(0, 3, 1, 2)
    ref3 = F.unfold(xr, 3, padding=1).view(G, C, 9, S * S).permute(0, 3, 2, 1).reshape(G * S * S, 9 * C)
    assert torch.equal(c3.float(), ref3)
    dc = rnd(G * S * S, 9 * C, seed=4).to(torch.bfloat16)
    dx = ops.col2im3x3(dc, G, S, S, C)
    refdx = F.fold(dc.float().view(G, S * S, 9, C).permute(0, 3, 2, 1).reshape(G, C * 9, S * S), (S, S), 3, padding=1)
    assert rel_err(dx.view(G, S, S, C), refdx.permute(0, 2, 3, 1)) < 1e-6


@pytest.mark.parametrize("M,N,K,rpg", [(8192, 192, 768, 4096), (8192, 768, 192, 4096), (800, 192, 768, 400), (512, 64, 64, 128)])
def test_gemm_grouped_weights(ops, M, N, K, rpg):
    """per-modality dense layers in one launch: rows of group g multiply w[g] and add bias[g] (whole-tile groups, and
    groups that do not end on a 128-row tile -> launched one by one); GELU + pre-activation and the x gelu'(aux) dgrad."""
    G = 2
    a = rnd(M, K, seed=1).to(torch.bfloat16)
    w = rnd(G, N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
    b = rnd(G, N, seed=3)
    grp = (torch.arange(M, device="cuda") // rpg) % G
    ref = torch.einsum("mk,mnk->mn", a.double(), w.double()[grp]) + b.double()[grp]
    out = ops.gemm_grouped(a, w, bias=b, rows_per_group=rpg, out_dtype=torch.float32)
    assert rel_err(out, ref) < 1e-5
    act, pre = ops.gemm_grouped(a, w, bias=b, rows_per_group=rpg, act=1, want_pre=True)
    assert rel_err(pre.float(), ref) < 5e-3 and rel_err(act.float(), F.gelu(ref)) < 5e-3
    aux = rnd(M, N, seed=4).to(torch.bfloat16)
    x = aux.double().requires_grad_()
    F.gelu(x).sum().backward()
    d = ops.gemm_grouped(a, w, rows_per_group=rpg, act=2, aux=aux, out_dtype=torch.float32)
    assert rel_err(d, (ref - b.double()[grp]) * x.grad) < 1e-4


# ------------------------------------------------------------------------------------------ attention
def _attn_reference(qkv, bias, rel_h, rel_w, G, S, heads, ws):
    """fp64 restatement through the oracle's window/rel-pos functions: padded tokens carry qkv = bias."""
    from oracle import mfnet_ref as R
    D = qkv.shape[1] // 3
    hd = D // heads
    t = qkv.view(G, S, S, 3 * D)
    if ws > 0:
        pad = (ws - S % ws) % ws
        tp = bias.view(1, 1, 1, -1).expand(G, S + pad, S + pad, 3 * D).clone()
        tp[:, :S, :S] = t
        Hp = S + pad
        win = tp.view(G, Hp // ws, ws, Hp // ws, ws, 3 * D).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, 3 * D)
        h = w = ws
    else:
        win = t.reshape(G, S * S, 3 * D)
        h = w = S
    q, k, v = R.split_heads(win, heads)
    o = R.attention_core(q, k, v, rel_h, rel_w, (h, w), hd ** -0.5)
    o = R.merge_heads(o, win.shape[0], heads, h, w)
    if ws > 0:
        o = R.window_unpartition(o, ws, (Hp, Hp), (S, S))
    return o.reshape(G * S * S, D)


def _rel_tables(rel_h, rel_w, k):
    """[rel_h ; rel_w] resized to 2k-1 rows each, zero-padded to a multiple of 32 rows (bf16)."""
    from ssrs_amd.vit import build_rel_table
    return build_rel_table(rel_h, rel_w, k, k)


@pytest.mark.parametrize("S,ws,heads,G,hd", [(16, 14, 2, 2, 64), (16, 0, 2, 2, 64), (20, 14, 3, 1, 64), (20, 0, 1, 2, 64), (16, 14, 12, 4, 64),
                                             (32, 0, 1, 1, 64), (64, 14, 1, 1, 64), (64, 0, 1, 1, 64),   # 2 and 4 slot groups; 1024-px grid
                                             (16, 14, 3, 2, 80), (16, 0, 2, 2, 80), (20, 14, 1, 1, 80), (32, 0, 1, 1, 80), (64, 14, 1, 1, 80),
                                             (64, 0, 1, 1, 80)])                                         # ViT-H head dim
def test_attention_fwd_bwd(ops, S, ws, heads, G, hd):
    D = heads * hd
    qkv = rnd(G * S * S, 3 * D, seed=1, scale=1.5).to(torch.bfloat16)
    bias = rnd(3 * D, seed=2)
    L = 27 if ws > 0 else 127
    rel_h = rnd(L, hd, seed=3, scale=0.2)
    rel_w = rnd(L, hd, seed=4, scale=0.2)
    k = ws if ws > 0 else S
    tbl, tblT, Lh, Lw = _rel_tables(rel_h, rel_w, k)
    assert tbl.shape[1] % 32 == 0 and tblT.shape[0] == hd
    out, lse = ops.attn_fwd(qkv, bias, tbl, G, S, S, heads, ws, Lh, Lw, hd ** -0.5)
    # reference on the same rounded operands (bf16 tables, bf16 padded-key bias), fp64
    rh = tbl[:Lh, :hd].double().cpu()
    rw = tbl[Lh:Lh + Lw, :hd].double().cpu()
    qd = qkv.double().cpu().requires_grad_()
    ref = _attn_reference(qd, bf16r(bias).double().cpu(), rh, rw, G, S, heads, ws)
    assert rel_err(out.float(), ref.detach()) < 1e-2
    dout = rnd(G * S * S, D, seed=5).to(torch.bfloat16)
    ref.backward(dout.double().cpu())
    dqkv = ops.attn_bwd(qkv, bias, tbl, tblT, out, dout, lse, G, S, S, heads, ws, Lh, Lw, hd ** -0.5)
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        e = rel_err(dqkv[:, sl].float(), qd.grad[:, sl])
        assert e < 2e-2, (name, e)


# ------------------------------------------------------------------------------------------ loss / sgd / merge
@pytest.mark.parametrize("C,ignore", [(6, -100), (2, 255)])
def test_ce_loss(ops, C, ignore):
    B, H, W = 3, 64, 80
    logits = rnd(B, C, H, W, seed=1, scale=4.0)
    tgt = torch.randint(0, C, (B, H, W), generator=torch.Generator().manual_seed(2)).cuda()
    if ignore == 255:
        tgt[:, :10, :20] = 255
    wt = (1 + 0.5 * rnd(C, seed=3)) if ignore != 255 else None
    acc, pred = ops.ce_loss_fwd(logits, tgt, wt, ignore, want_pred=True)
    ld = logits.double().requires_grad_()
    ref = F.cross_entropy(ld.permute(0, 2, 3, 1).reshape(-1, C), tgt.view(-1),
                          wt.double() if wt is not None else None, ignore_index=ignore)
    ref.backward()
    assert abs((acc[0] / acc[1]).item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
    assert torch.equal(pred.long(), logits.argmax(1))
    d = ops.ce_loss_bwd(logits, tgt, acc, wt, ignore, gscale=1.0)
    assert rel_err(d, ld.grad) < 1e-4


def test_sgd_flat(ops):
    n = 100003
    p = rnd(n, seed=1)
    g = rnd(n, seed=2)
    pr = torch.nn.Parameter(p.clone())
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=5e-4)
    buf = torch.zeros_like(p)
    for step in range(3):
        pr.grad = g.clone() * (step + 1)
        opt.step()
        ops.sgd_flat(p, g * (step + 1), buf, 0.01, 0.9, 5e-4)
    assert rel_err(p, pr.data) < 1e-6


def test_tile_accumulate_argmax_minmax(ops):
    C, H, W, h, w = 3, 300, 280, 64, 64
    pred = torch.zeros(C, H, W, device="cuda")
    coords = torch.tensor([[0, 0], [32, 16], [236, 216], [236, 216], [100, 100]], dtype=torch.int32, device="cuda")
    logits = rnd(5, C, h, w, seed=1)
    ops.tile_accumulate(pred, logits, coords, atomic=True)
    ref = torch.zeros(C, H, W, dtype=torch.float64)
    for i, (x, y) in enumerate(coords.tolist()):
        ref[:, x:x + h, y:y + w] += logits[i].double().cpu()
    assert rel_err(pred, ref) < 1e-6
    assert torch.equal(ops.argmax_chw(pred).long().cpu(), pred.cpu().argmax(0))
    x = rnd(1000, 999, seed=3, scale=50.0) + 7
    mm = ops.minmax(x)
    assert mm[0].item() == x.min().item() and mm[1].item() == x.max().item()
    y = ops.scale_shift(x, mm, 0)
    assert rel_err(y, (x - x.min()) / (x.max() - x.min())) < 1e-6


@pytest.mark.parametrize("M,N,K", [(65536, 64, 576), (16384, 256, 1024), (4100, 32, 64), (300, 192, 768), (1000, 24, 72), (333, 40, 136), (65536, 64, 576)])
def test_gemm_tn_split_k(ops, M, N, K):
    """weight-gradient product dW = dY^T X on the row-major activations (transposing LDS reads) + split reduction;
    ragged M / N / K included."""
    dy = rnd(M, N, seed=1).to(torch.bfloat16)
    x = rnd(M, K, seed=2).to(torch.bfloat16)
    got = ops.gemm_tn(dy, x)
    ref = dy.double().t() @ x.double()
    assert got.shape == (N, K)
    assert rel_err(got, ref) < 1e-4
    got2, db = ops.gemm_tn(dy, x, lazy=True, want_bias=True)       # bias gradient (column sums of dy) from the same launch
    got2 = got2.materialize() if isinstance(got2, ops.SplitSum) else got2
    assert rel_err(got2, got) < 1e-6                                # (the two reductions sum the split tiles in different orders)
    assert db.shape == (N,) and rel_err(db.materialize(), dy.double().sum(0)) < 1e-5
    if M % 2 == 0:
        gw, gb = ops.gemm_tn_grouped(dy, x, 2, want_bias=True)
        h = M // 2
        assert rel_err(gw.materialize(), torch.stack([dy[:h].double().t() @ x[:h].double(), dy[h:].double().t() @ x[h:].double()])) < 1e-4
        assert rel_err(gb.materialize(), torch.stack([dy[:h].double().sum(0), dy[h:].double().sum(0)])) < 1e-5
        assert rel_err(gb[1].materialize(), dy[h:].double().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------ batched strided copies
def test_gather_segments(ops):
    """mfnet_gather_segments: permuted / sliced / broadcast sources, sliced destinations, fp32 assign and
    accumulate, bf16 assign, and more segments than fit one launch (48)."""
    base = rnd(7, 5, 3, 4, seed=1)
    mats = [rnd(33, 17, seed=10 + i) for i in range(60)]
    pairs, refs = [], []
    d0 = torch.zeros(4, 3, 5, 7, device="cuda")
    pairs.append((d0, base.permute(3, 2, 1, 0))); refs.append((d0, base.permute(3, 2, 1, 0).clone()))
    d1 = torch.zeros(40, 20, device="cuda")
    pairs.append((d1[3:36, 2:19], mats[0]))
    r1 = torch.zeros(40, 20, device="cuda"); r1[3:36, 2:19] = mats[0]
    refs.append((d1, r1))
    d2 = torch.zeros(4, 17, device="cuda")
    pairs.append((d2, mats[1][5].expand(4, 17))); refs.append((d2, mats[1][5].expand(4, 17).clone()))
    for i in range(2, 60):
        d = torch.zeros(17, 33, device="cuda")
        pairs.append((d, mats[i].t())); refs.append((d, mats[i].t().clone()))
    ops.gather(pairs)
    for d, r in refs:
        assert torch.equal(d, r)
    ops.gather(pairs, accumulate=True)
    for d, r in refs:
        assert torch.equal(d, 2 * r)
    db = torch.zeros(17, 33, dtype=torch.bfloat16, device="cuda")
    ops.gather([(db, mats[3].t())])
    assert torch.equal(db, mats[3].t().to(torch.bfloat16))


def test_split_sums_are_folded_into_the_gather(ops):
    """SplitSum sources: the partial tiles of the split weight-gradient reductions are summed (x scale) by the gather
    launch that delivers them; lazy skinny_tn / gemm_tn results equal the eager ones bit for bit where the order is
    the same, and through views / slices / transposes."""
    part = rnd(5, 12, 20, seed=3)
    ss = ops.SplitSum(part, 0.5)
    d = torch.zeros(20, 12, device="cuda")
    ops.gather([(d, ss.t())])
    assert rel_err(d, 0.5 * part.sum(0).t()) < 1e-6
    d2 = torch.ones(3, 4, 6, device="cuda")
    ops.gather([(d2[:, :3], ss[3:6, :18].view(3, 3, 6))], accumulate=True)
    ref2 = torch.ones(3, 4, 6, device="cuda")
    ref2[:, :3] += 0.5 * part.sum(0)[3:6, :18].reshape(3, 3, 6)
    assert rel_err(d2, ref2) < 1e-6
    assert rel_err(ss.materialize(), 0.5 * part.sum(0)) < 1e-6
    # lazy weight-gradient products
    y = rnd(4096, 768, seed=4).to(torch.bfloat16)
    u = rnd(4096, 8, seed=5)
    for tr in (False, True):
        e = ops.skinny_tn(y, u, tr, 0.25, n_groups=2)
        lz = ops.skinny_tn(y, u, tr, 0.25, n_groups=2, lazy=True)
        assert tuple(lz.shape) == tuple(e.shape)
        assert rel_err(lz.materialize(), e) < 1e-6
        assert rel_err(lz[1].materialize(), e[1]) < 1e-6
    e1 = ops.skinny_tn(y, u, False)
    l1 = ops.skinny_tn(y, u, False, lazy=True)
    assert rel_err(l1[:, :4].materialize(), e1[:, :4]) < 1e-6
    dy = rnd(65536, 64, seed=6).to(torch.bfloat16)
    x = rnd(65536, 576, seed=7).to(torch.bfloat16)
    e2 = ops.gemm_tn(dy, x)
    l2 = ops.gemm_tn(dy, x, lazy=True)
    assert isinstance(l2, ops.SplitSum) and rel_err(l2.materialize(), e2) < 1e-6
    assert rel_err(l2.view(64, 3, 3, 64).permute(0, 3, 1, 2).materialize(), e2.view(64, 3, 3, 64).permute(0, 3, 1, 2)) < 1e-6
    # two groups of rows in one launch (the per-modality adapter weight gradients)
    dyg = rnd(8192, 192, seed=8).to(torch.bfloat16)
    xg = rnd(8192, 768, seed=9).to(torch.bfloat16)
    lg = ops.gemm_tn_grouped(dyg, xg, 2)
    assert tuple(lg.shape) == (2, 192, 768)
    for gi in range(2):
        ref = dyg[gi * 4096:(gi + 1) * 4096].double().t() @ xg[gi * 4096:(gi + 1) * 4096].double()
        assert rel_err(lg[gi].materialize(), ref) < 1e-5


@pytest.mark.parametrize("M,N", [(4096, 768), (4096, 192), (16384, 256), (65536, 64), (1000, 3072), (77, 1280)])
def test_bias_grad(ops, M, N):
    """column sums of a bf16 gradient: widths that take the deterministic reduction (multiples of 4 up to 1024, the ViT
    adapters' 192 / 768 included) and widths that take the atomic kernel (3072, 1280)."""
    dy = rnd(M, N, seed=11).to(torch.bfloat16)
    out = ops.bias_grad(dy)
    assert tuple(out.shape) == (N,)
    assert rel_err(out, dy.double().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------ column reductions
@pytest.mark.parametrize("G,P,C,dt", [(1, 65536, 64, torch.float32), (1, 16384, 256, torch.bfloat16), (16, 4096, 64, torch.float32),
                                      (3, 777, 256, torch.float32), (1, 100, 1024, torch.float32), (2, 9, 8, torch.bfloat16),
                                      (1, 4096, 768, torch.bfloat16), (2, 500, 192, torch.float32), (1, 300, 1000, torch.float32),
                                      (1, 2048, 192, torch.bfloat16), (1, 64, 12, torch.float32)])   # widths whose quarter does not divide 256
def test_colsum_coldot_deterministic(ops, G, P, C, dt):
    """BatchNorm statistics / SE pools: values against float64, and bit-identical results run after run (the
    finish is a fixed-order sum of per-workgroup slabs, not float atomics)."""
    x = rnd(G * P, C, seed=1).to(dt)
    y = rnd(G * P, C, seed=2).to(dt)
    s, sq = ops.colsum(x, G, P, want_sq=True)
    d = ops.coldot(x, y, G, P)
    xd, yd = x.double().view(G, P, C), y.double().view(G, P, C)
    assert rel_err(s, xd.sum(1)) < 1e-5 and rel_err(sq, (xd * xd).sum(1)) < 1e-5 and rel_err(d, (xd * yd).sum(1)) < 1e-5
    for _ in range(5):
        s2, sq2 = ops.colsum(x, G, P, want_sq=True)
        assert torch.equal(s, s2) and torch.equal(sq, sq2) and torch.equal(d, ops.coldot(x, y, G, P))


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,Cr,relu6,bias", [(1, 256, 16, False, True), (16, 256, 16, False, True), (5, 64, 4, True, False),
                                               (20, 256, 16, False, True), (33, 64, 4, True, False)])
def test_se_gate_forward_backward_vs_float64_autograd(B, C, Cr, relu6, bias):
    """squeeze-excite gate (unetformer_mmsam.py:251-266, 333-337), SE (C=256, Cr=16, biases, ReLU) and FRH (C=64, Cr=4,
    ReLU6) shapes over several batch sizes, against torch autograd in float64 (fp32 tolerance 1e-5 relative to the largest entry)."""
    import torch
    from ssrs_amd import ops
    gen = torch.Generator().manual_seed(B * 1000 + C)
    P = 64
    pooled = (torch.randn(B, C, generator=gen) * P).cuda()
    W1 = (torch.randn(Cr, C, generator=gen) * 0.2).cuda()
    W2 = (torch.randn(C, Cr, generator=gen) * 0.2).cuda()
    b1 = (torch.randn(Cr, generator=gen) * 0.1).cuda() if bias else None
    b2 = (torch.randn(C, generator=gen) * 0.1).cuda() if bias else None
    dg = torch.randn(B, C, generator=gen).cuda()
    z, g = ops.se_gate_fwd(pooled, P, W1, b1, W2, b2, relu6)
    dpool, dW1, db1, dW2, db2 = ops.se_gate_bwd(dg, pooled, P, W1, W2, relu6, z, g, bias)

    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (pooled, W1, b1, W2, b2)]
    p64, w1, bb1, w2, bb2 = leaves
    h = (p64 / P) @ w1.t() + (bb1 if bias else 0.0)
    h = torch.clamp(h, 0.0, 6.0) if relu6 else torch.relu(h)
    gate = torch.sigmoid(h @ w2.t() + (bb2 if bias else 0.0))
    gate.backward(dg.double())

    def close(a, b):
        assert (a.double() - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    close(g, gate.detach())
    close(dpool, p64.grad)           # d/d(pooled sum) = d/d(mean) / P
    close(dW1, w1.grad)
    close(dW2, w2.grad)
    if bias:
        close(db1, bb1.grad)
        close(db2, bb2.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C,act", [(4096, 64, 1), (1000, 256, 0), (64, 1024, 2)])
def test_bn_train_apply_equals_finalize_then_apply(rows, C, act):
    """mfnet_bn_train_apply (one launch) against mfnet_bn_finalize + mfnet_bn_apply: outputs, saved statistics and the
    running-statistics update (nn.BatchNorm2d, momentum 0.1, unbiased running variance) bit for bit."""
    import torch
    from ssrs_amd import ops
    gen = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=gen) * 2 + 0.5).cuda()
    gamma = torch.rand(C, generator=gen).cuda() + 0.5
    beta = torch.randn(C, generator=gen).cuda()
    rm0, rv0 = torch.randn(C, generator=gen).cuda(), torch.rand(C, generator=gen).cuda() + 0.5

    rm_a, rv_a, nbt_a = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device="cuda")
    y_a, y2_a, mean_a, rstd_a = ops.bn_forward(x, gamma, beta, rm_a, rv_a, nbt_a, True, act=act, out2_dtype=torch.bfloat16)

    rm_b, rv_b, nbt_b = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device="cuda")
    s, sq = ops.colsum(x, 1, rows, want_sq=True)
    st = torch.empty((2, C), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ops.call("mfnet_bn_finalize", s.data_ptr(), sq.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), rm_b.data_ptr(), rv_b.data_ptr(),
             nbt_b.data_ptr(), C, rows, 1e-5, 0.1, 1, stream)
    y_b = torch.empty_like(y_a)
    y2_b = torch.empty_like(y2_a)
    ops.call("mfnet_bn_apply", x.data_ptr(), 0, st[0].data_ptr(), st[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, None, 0,
             y_b.data_ptr(), 0, y2_b.data_ptr(), 1, rows, C, stream)
    torch.cuda.synchronize()
    for a, b in ((y_a, y_b), (y2_a, y2_b), (mean_a, st[0]), (rstd_a, st[1]), (rm_a, rm_b), (rv_a, rv_b), (nbt_a, nbt_b)):
        assert torch.equal(a, b)
    assert int(nbt_a) == 1
    ref = torch.nn.functional.batch_norm(x.double(), None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    if act == 0:
        assert (y_a.double() - ref).abs().max().item() < 1e-4


# ------------------------------------------------------------------------------------------ depthwise conv
@pytest.mark.parametrize("B,H,W,C,k,ext,bf", [(2, 8, 8, 64, 8, True, False), (2, 16, 12, 64, 8, True, True), (1, 32, 32, 64, 8, True, False),
                                              (2, 16, 16, 64, 3, False, False), (1, 9, 7, 32, 3, False, True), (1, 10, 10, 16, 5, False, False)])
def test_depthwise_conv_fwd_bwd_vs_float64(ops, B, H, W, C, k, ext, bf):
    """depthwise k x k conv on channels-last maps (unetformer_mmsam.py:62-72 SeparableConvBN's first conv; ext = the reflected
    bottom / right row of GlobalLocalAttention.pad_out, :192-196): forward, data gradient, weight and bias gradients against
    F.conv2d(groups=C) in float64; k = 3 and 8 take the row-batched kernels, k = 5 the generic one."""
    x = rnd(B, H, W, C, seed=1)
    w = rnd(C, 1, k, k, seed=2) * 0.2
    bias = rnd(C, seed=3)
    dy = rnd(B, H, W, C, seed=4)
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    bd = bias.double().requires_grad_(True)
    pad = (k - 1) // 2
    xin = F.pad(xd, (0, 1, 0, 1), mode="reflect") if ext else xd
    ref = F.conv2d(xin, wd, bd, padding=pad, groups=C)[:, :, :H, :W]
    dyd = (dy.bfloat16() if bf else dy).double().permute(0, 3, 1, 2)
    ref.backward(dyd)
    xc, wc = x.cuda(), w.cuda().reshape(C, k * k).contiguous()
    y = ops.dwconv_fwd(xc, wc, bias.cuda(), B, H, W, k, ext)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-5
    dyc = dy.cuda().bfloat16() if bf else dy.cuda()
    dx, dw, db = ops.dwconv_bwd(dyc, xc, wc, B, H, W, k, ext, want_bias=True)
    assert rel_err(dx.permute(0, 3, 1, 2), xd.grad) < 1e-5
    assert rel_err(dw.materialize().view(C, 1, k, k), wd.grad) < 1e-5
    assert rel_err(db.materialize(), bd.grad) < 1e-5


# ------------------------------------------------------------------------------------------ implicit-GEMM 3x3 convolution
@pytest.mark.parametrize("G,H,W,C,N", [(2, 16, 16, 64, 64), (1, 8, 8, 64, 64), (3, 12, 20, 128, 48), (2, 16, 16, 256, 256), (1, 5, 7, 64, 16),
                                       (16, 64, 64, 64, 64)])
def test_conv3x3_implicit_gemm_fwd_dgrad_wgrad(ops, G, H, W, C, N):
    """3x3 / pad 1 convs of the decoder and the neck without a column tensor (unetformer_mmsam.py:43-60, image_encoder.py:112-118):
    forward, data gradient (same kernel, flipped + transposed weights) and weight gradient against F.conv2d in float64 on the
    bf16-rounded operands; partial tiles (M % 128, N < 128), several k blocks per tap (C = 128, 256) and image borders included."""
    x = bf16r(rnd(G, H, W, C, seed=1))
    w = bf16r(rnd(N, C, 3, 3, seed=2) * 0.1)
    dy = bf16r(rnd(G, H, W, N, seed=3))
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, padding=1)
    ref.backward(dy.double().permute(0, 3, 1, 2))
    xb = x.cuda().bfloat16().view(-1, C)
    wb = w.cuda().permute(0, 2, 3, 1).reshape(N, 9 * C).bfloat16().contiguous()
    y = ops.conv3x3(xb, wb, G, H, W)
    assert rel_err(y.view(G, H, W, N).permute(0, 3, 1, 2), ref) < 1e-5
    yb = ops.conv3x3(xb, wb, G, H, W, out_dtype=torch.bfloat16)
    assert rel_err(yb.float().view(G, H, W, N).permute(0, 3, 1, 2), ref) < 5e-3
    dyb = dy.cuda().bfloat16().view(-1, N)
    wT = w.cuda().flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * N).bfloat16().contiguous()
    dx = ops.conv3x3(dyb, wT, G, H, W) if N % 64 == 0 else None
    if dx is not None:
        assert rel_err(dx.view(G, H, W, C).permute(0, 3, 1, 2), xd.grad) < 1e-5
    dW, db = ops.conv3x3_wgrad(dyb, xb, G, H, W, want_bias=True)
    dW = dW.materialize() if isinstance(dW, ops.SplitSum) else dW
    assert rel_err(dW.view(N, 3, 3, C).permute(0, 3, 1, 2), wd.grad) < 1e-4
    assert rel_err(db.materialize(), dy.double().sum((0, 1, 2))) < 1e-5
    # the column-tensor path it replaces gives the same numbers
    cols = ops.im2col3x3(xb.contiguous(), G, H, W)
    assert rel_err(ops.gemm(cols, wb, out_dtype=torch.float32), y) < 1e-6


def test_conv3x3_rejects_bad_geometry(ops):
    from ssrs_amd._lib import MfnetError
    x = torch.zeros(64, 32, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(16, 9 * 32, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(MfnetError):
        ops.conv3x3(x, w, 1, 8, 8)                     # channels not a multiple of 64


@pytest.mark.parametrize("n,h,w", [(32, 256, 256), (3, 10, 10), (2, 7, 1), (1, 300, 300)])
def test_tile_minmax_norm(ops, n, h, w):
    """per-tile DSM min-max (inference.py:547-555; multiscale_dataset.py:205-208 with keep_flat): the register-resident kernel
    (tiles up to 65536 values) and the two-pass one give the bits of the fp32 statement; a flat tile becomes 0 / stays."""
    x = rnd(n, h, w, seed=11, scale=30.0)
    x[0] = 4.25                                              # a flat tile
    for keep in (False, True):
        y = x.clone()
        ops.tile_minmax_norm(y, keep_flat=keep)
        lo = x.amin(dim=(1, 2), keepdim=True)
        rng = x.amax(dim=(1, 2), keepdim=True) - lo
        ref = torch.where(rng > 0, (x - lo) / rng, torch.zeros_like(x))
        if keep:
            ref[0] = x[0]
        assert torch.equal(y, ref)


def test_gemm_tn_batch_equals_one_by_one(ops):
    """mfnet_gemm_tn_batch (the step's weight-gradient products collected and launched side by side) against the one-problem
    entry points: plain, grouped and 3x3-conv problems of very different sizes in one batch, more problems than one launch
    holds (24) - bit-identical partial tiles and bias rows."""
    from ssrs_amd import ops as O
    probs = []
    for i, (M, N, K) in enumerate([(65536, 64, 64), (1024, 64, 256), (4096, 256, 64), (16384, 1024, 256), (1000, 72, 40)] * 6):
        probs.append(("plain", rnd(M, N, seed=100 + i).to(torch.bfloat16), rnd(M, K, seed=200 + i).to(torch.bfloat16), i % 2 == 0))
    probs.append(("grouped", rnd(8192, 192, seed=7).to(torch.bfloat16), rnd(8192, 768, seed=8).to(torch.bfloat16), True))
    probs.append(("conv", rnd(2 * 16 * 16, 64, seed=9).to(torch.bfloat16), rnd(2 * 16 * 16, 64, seed=10).to(torch.bfloat16), True))

    def run():
        outs = []
        for kind, dy, x, wb in probs:
            if kind == "plain":
                r = O.gemm_tn(dy, x, lazy=True, want_bias=wb)
            elif kind == "grouped":
                r = O.gemm_tn_grouped(dy, x, 2, want_bias=wb)
            else:
                r = O.conv3x3_wgrad(dy, x, 2, 16, 16, want_bias=wb)
            outs.append(r if isinstance(r, tuple) else (r,))
        return outs
    assert O.TN_QUEUE is None
    ref = run()
    O.TN_QUEUE = []
    try:
        got = run()
        assert len(O.TN_QUEUE) == len(probs)
        O.tn_flush()
        assert O.TN_QUEUE == []
    finally:
        O.TN_QUEUE = None
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        for u, v in zip(a, b):
            pu = u.part if isinstance(u, O.SplitSum) else u
            pv = v.part if isinstance(v, O.SplitSum) else v
            assert torch.equal(pu, pv)


def test_lora_side_batch_equals_one_by_one(ops):
    """mfnet_lora_side_batch (the dA = x^T . da products of a step, collected) against mfnet_lora_side, both ranks, bit-identical."""
    from ssrs_amd import ops as O
    probs = [(rnd(8192, C, seed=30 + i).to(torch.bfloat16), rnd(8192, R, seed=60 + i)) for i, (C, R) in enumerate([(768, 4), (3072, 4), (768, 8), (128, 8), (3072, 4)] * 6)]
    ref = [O.lora_side(y, U=u, n_groups=2)[0].part.clone() for y, u in probs]
    O.TN_QUEUE = []
    try:
        got = [O.lora_side(y, U=u, n_groups=2)[0] for y, u in probs]
        assert len(O.TN_QUEUE) == len(probs)
        O.tn_flush()
    finally:
        O.TN_QUEUE = None
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b.part)


def test_dwconv_bwd_weight_batch_matches_one_by_one(ops):
    """mfnet_dwconv_bwd_weight_batch (collected depthwise weight gradients) against the one-problem entry point: all three kernel
    variants in one batch. Float atomics: equal up to the summation order."""
    from ssrs_amd import ops as O
    cases = [(16, 8, 8, True), (16, 32, 8, True), (16, 64, 3, False), (2, 20, 3, True), (16, 16, 8, True)]
    data = [(rnd(B * S * S, 64, seed=40 + i), rnd(B * S * S, 64, seed=50 + i), rnd(64, k * k, seed=70 + i), B, S, k, ext) for i, (B, S, k, ext) in enumerate(cases)]
    ref = [O.dwconv_bwd(dy, x, w, B, S, S, k, ext, want_bias=True) for dy, x, w, B, S, k, ext in data]
    ref = [(r[1].materialize(), r[2].materialize()) for r in ref]
    O.TN_QUEUE = []
    try:
        got = [O.dwconv_bwd(dy, x, w, B, S, S, k, ext, want_bias=True) for dy, x, w, B, S, k, ext in data]
        assert len(O.TN_QUEUE) == len(cases)
        O.tn_flush()
        got = [(r[1].materialize(), r[2].materialize()) for r in got]
    finally:
        O.TN_QUEUE = None
    for (a, b), (c, d) in zip(ref, got):
        assert rel_err(c, a) < 1e-5 and rel_err(d, b) < 1e-5
