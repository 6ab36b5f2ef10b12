"""fp32-accurate forward of MFNet (eval / inference): `UNetFormer(..., precision="fp32")` or `model.precision = "fp32"`.

The reference computes everything in fp32 (unetformer_mmsam.py:533-563, lora_block.py:169-196); the default HIP path runs
its contractions on bf16 MFMA (2e-2 on the logits). This path keeps every activation in fp32 and runs every contraction on
the fp32 MFMA (csrc/fp32.hip: `mfnet_gemm_f32`, a k-ordered fp32 fma chain) and the attention in fp32 (`mfnet_attn_f32`,
`mfnet_gla_attn_fwd_f32`), so the logits agree with the fp32 reference to 1e-4 relative. It reuses the fp32 kernels the
training path already has for everything that is not a contraction (LayerNorm, BatchNorm, SE gates, pooling, depthwise
convolutions, the up-sampling fusions). Forward only, no saved tensors: fp32 TRAINING runs the ordinary module path in
`ops.precision("fp32")` mode (unetformer_mmsam.UNetFormer.forward), which owns the backward passes.

Same module objects, same parameters - the functions below only read `model`'s tensors."""
from __future__ import annotations

import torch

from . import decoder_ops as D
from . import ops
from ._lib import call
from .ops import ACT_GELU, ACT_NONE, ACT_RELU6
from .vit import _cached, _resize_rel

F32 = torch.float32


# ---------------------------------------------------------------------------------------------- kernels
gemm32 = ops.gemm_f32
attn32 = ops.attn_f32


def im2col_patch32(rgb, dsm, patch=16):
    ops._chk(rgb, dsm)
    B, _, H, W = rgb.shape
    ych = 1 if dsm.dim() == 3 else int(dsm.shape[1])
    if tuple(dsm.shape) != ((B, H, W) if dsm.dim() == 3 else (B, ych, H, W)) or rgb.shape[1] != 3 or ych not in (1, 3):
        raise ValueError(f"im2col_patch: rgb {tuple(rgb.shape)} and second modality {tuple(dsm.shape)} do not match")
    cols = torch.empty((2 * B * (H // patch) * (W // patch), 3 * patch * patch), dtype=F32, device=rgb.device)
    call("mfnet_im2col_patch_f32", rgb.data_ptr(), dsm.data_ptr(), ych, cols.data_ptr(), B, H, W, patch, ops._stream(rgb))
    return cols


def conv3x3_32(x, w, B, H, W):
    """3x3 / pad 1 convolution of channels-last fp32 rows x [B*H*W, I] with weight [O, I, 3, 3] -> [B*H*W, O]."""
    I = x.shape[1]
    cols = torch.empty((B * H * W, 9 * I), dtype=F32, device=x.device)
    call("mfnet_im2col3x3_f32", x.data_ptr(), cols.data_ptr(), B, H, W, I, ops._stream(x))
    return gemm32(cols, w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], 9 * I).contiguous())      # k = tap*I + i


def _w2(w):
    return w.detach().reshape(w.shape[0], -1)


def _ln(x, ln):
    return ops.layernorm_fwd(x, ln.weight, ln.bias, ln.eps, out_dtype=F32, want_stats=False)[0]


def _bn(x, bn, act=ACT_NONE):
    return ops.bn_forward(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, None, False, bn.eps, 0.1, act, None, 0, F32, None)[0]


# ---------------------------------------------------------------------------------------------- encoder
def _rel_tables32(attn, kh, kw):
    """get_rel_pos's resized tables (lora_block.py:260-271) as fp32 [2k-1, hd] matrices; frozen, cached per window size."""
    return _cached(("rel32", kh, kw), (attn.rel_h, attn.rel_w),
                   lambda: (_resize_rel(attn.rel_h.detach().float(), 2 * kh - 1).contiguous(),
                            _resize_rel(attn.rel_w.detach().float(), 2 * kw - 1).contiguous()))


def _attention32(qkv, attn, G, Sh, Sw, heads, ws):
    kh, kw = (ws, ws) if ws > 0 else (Sh, Sw)
    th, tw = _rel_tables32(attn, kh, kw)
    hd = qkv.shape[1] // (3 * heads)
    return attn32(qkv, attn.qkv.bias.detach(), th, tw, G, Sh, Sw, heads, ws, hd ** -0.5)


def _lora_mix_linear32(x, lin, half, act=0, residual=None):
    """lora.Linear.forward on stacked rows (layers.py:149-165): dense + [wx*lx + (1-wx)*ly | wy*ly + (1-wy)*lx]."""
    r = lin.lora_Ax.shape[0]
    if r % 4:
        raise NotImplementedError("fp32 path: LoRA rank must be a multiple of 4 (the fp32 MFMA consumes k in fours)")
    a = torch.empty((x.shape[0], r), dtype=F32, device=x.device)
    gemm32(x[:half], lin.lora_Ax.detach(), out=a[:half])
    gemm32(x[half:], lin.lora_Ay.detach(), out=a[half:])
    U = ops.lora_mix_fwd(a, lin.lora_wx.detach(), lin.lora_wy.detach())                       # [M, 2r]
    V = torch.cat([lin.lora_Bx.detach(), lin.lora_By.detach()], 1).contiguous()             # [N, 2r]
    pre = gemm32(x, lin.weight.detach(), bias=lin.bias.detach(), residual=residual)
    gemm32(U, V, out=pre, accumulate=True)
    return ops.gelu_fwd(pre) if act else pre


def lora_block32(blk, h, B, Sh, Sw):
    """LoraBlock.forward (lora_block.py:64-87) on the stacked token matrix h [2*B*Sh*Sw, D] fp32."""
    M, Dm = h.shape
    half = M // 2
    attn = blk.attn
    q = attn.qkv
    r = q.r
    if r % 4:
        raise NotImplementedError("fp32 path: LoRA rank must be a multiple of 4 (the fp32 MFMA consumes k in fours)")
    ln = _ln(h, blk.norm1)
    qkv = gemm32(ln, q.weight.detach(), bias=q.bias.detach())
    for g, (A, Bm) in enumerate(((q.lora_Ax, q.lora_Bx), (q.lora_Ay, q.lora_By))):           # per-modality q and v updates
        rows = slice(g * half, (g + 1) * half)
        a = gemm32(ln[rows], A.detach())                                                      # [half, 2r]: r for q, r for v
        gemm32(a[:, :r], Bm.detach()[:Dm], alpha=q.scaling, out=qkv[rows, :Dm], accumulate=True)
        gemm32(a[:, r:], Bm.detach()[Dm:], alpha=q.scaling, out=qkv[rows, 2 * Dm:], accumulate=True)
    o = _attention32(qkv, attn, 2 * B, Sh, Sw, blk.num_heads, blk.window_size)
    h1 = gemm32(o, attn.proj.weight.detach(), bias=attn.proj.bias.detach(), residual=h)
    ln2 = _ln(h1, blk.norm2)
    act = _lora_mix_linear32(ln2, blk.mlp.lin1, half, act=1)
    return _lora_mix_linear32(act, blk.mlp.lin2, half, residual=h1)


def _adapter32(ad, x, out=None, residual=None, accumulate=False):
    hid = gemm32(x, ad.D_fc1.weight.detach(), bias=ad.D_fc1.bias.detach(), act=1)
    return gemm32(hid, ad.D_fc2.weight.detach(), bias=ad.D_fc2.bias.detach(), residual=residual, out=out, accumulate=accumulate)


def adapter_block32(blk, h, B, Sh, Sw):
    """AdapterFusionBlock.forward (adapter_fusionblock.py:76-114) on the stacked token matrix."""
    M, Dm = h.shape
    half = M // 2
    attn = blk.attn
    ln = _ln(h, blk.norm1)
    qkv = gemm32(ln, attn.qkv.weight.detach(), bias=attn.qkv.bias.detach())
    o = _attention32(qkv, attn, 2 * B, Sh, Sw, blk.num_heads, blk.window_size)
    po = gemm32(o, attn.proj.weight.detach(), bias=attn.proj.bias.detach())
    h1 = h.clone()
    for rows, ad in ((slice(0, half), blk.Img_Adapter), (slice(half, M), blk.DSM_Adapter)):   # skip adapters: po + ad(po), + h
        _adapter32(ad, po[rows], out=h1[rows], residual=po[rows], accumulate=True)
    ln2 = _ln(h1, blk.norm2)
    mlp = blk.mlp
    hid = gemm32(ln2, mlp.lin1.weight.detach(), bias=mlp.lin1.bias.detach(), act=1)
    hm = gemm32(hid, mlp.lin2.weight.detach(), bias=mlp.lin2.bias.detach(), residual=h1)
    ada = torch.empty((M, Dm), dtype=F32, device=h.device)
    _adapter32(blk.MLPx_Adapter, ln2[:half], out=ada[:half])
    _adapter32(blk.MLPy_Adapter, ln2[half:], out=ada[half:])
    return ops.adapter_mix(hm, ada, blk.wx_Adapter.detach(), blk.wy_Adapter.detach())


def encoder32(enc, rgb, dsm):
    """ImageEncoderViT (image_encoder.py:122-144) -> [2B, Sh, Sw, 256] channels-last fp32 (RGB images first)."""
    from .vit import LoraBlock
    B, _, H, W = rgb.shape
    P = enc.patch_size
    Sh, Sw = H // P, W // P
    cols = im2col_patch32(rgb.float().contiguous(), dsm.float().contiguous(), P)
    pe = enc.patch_embed.proj
    pos = enc._pos(Sh, Sw) if enc.pos_embed is not None else None
    h = gemm32(cols, _w2(pe.weight), bias=pe.bias.detach(), residual=pos, res_row_mod=Sh * Sw if pos is not None else 0)
    for blk in enc.blocks:
        h = (lora_block32 if isinstance(blk, LoraBlock) else adapter_block32)(blk, h, B, Sh, Sw)
    neck = enc.neck
    C = neck[0].weight.shape[0]
    n1 = gemm32(h, _w2(neck[0].weight))
    l1 = ops.layernorm_fwd(n1, neck[1].weight, neck[1].bias, neck[1].eps, out_dtype=F32, want_stats=False)[0]
    n2 = conv3x3_32(l1, neck[2].weight, 2 * B, Sh, Sw)
    out = ops.layernorm_fwd(n2, neck[3].weight, neck[3].bias, neck[3].eps, out_dtype=F32, want_stats=False)[0]
    return out.view(2 * B, Sh, Sw, C)


# ---------------------------------------------------------------------------------------------- FPN + fusion
def _ct32(x, ct):
    """ConvTranspose2d(k=2, s=2) as a GEMM with N = 4*Co; output row m holds the 4 sub-pixels (dy, dx) of input pixel m
    (the un-shuffled order of decoder_ops.py)."""
    w = ct.weight.detach()
    Ci, Co = w.shape[:2]
    return gemm32(x, w.permute(2, 3, 1, 0).reshape(4 * Co, Ci).contiguous(), bias=ct.bias.detach().repeat(4)).view(-1, Co)


def fpn_fusion32(model, deep):
    """fpn1..4{x,y} + fusion1..4 (unetformer_mmsam.py:494-516, 545-556) -> res1 (perm 2), res2 (perm 1), res3, res4 fp32."""
    twoB, h, w, C = deep.shape
    B = twoB // 2
    M3 = B * h * w
    d2 = deep.reshape(2 * M3, C)
    r = {}
    for mi, m in enumerate("xy"):
        f1, f2 = getattr(model, f"fpn1{m}"), getattr(model, f"fpn2{m}")
        dm = d2[mi * M3:(mi + 1) * M3]
        t1 = _ct32(dm, f1[0])
        n1 = _ln(t1, f1[1].ln) if m == "x" else _bn(t1, f1[1])
        r1 = _ct32(ops.act(n1, ACT_GELU), f1[3])
        r4, _ = ops.maxpool2_fwd(dm.contiguous(), B, h, w)
        r[m] = (r1, _ct32(dm, f2[0]), dm.contiguous(), r4)
    P = (16 * h * w, 4 * h * w, h * w, (h // 2) * (w // 2))
    outs = []
    for i in range(4):
        fu = getattr(model, f"fusion{i + 1}")
        gx = D.se_gates(fu.se_rgb, r["x"][i], B, P[i])[2]
        gy = D.se_gates(fu.se_depth, r["y"][i], B, P[i])[2]
        outs.append(ops.se_scale_sum(r["x"][i], r["y"][i], gx, gy, B, P[i], out_dtype=F32))
    return outs


# ---------------------------------------------------------------------------------------------- decoder
def _block32(blk, x, B, H, W):
    """Block = x + GLA(BN(x)); + Mlp(BN(.)) (unetformer_mmsam.py:86-217), eval-mode BatchNorm."""
    a = blk.attn
    heads, ws = a.num_heads, a.ws
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    if Hp - H >= H or Wp - W >= W:
        raise ValueError(f"GlobalLocalAttention: a {H}x{W} map cannot be reflect-padded to the {ws}x{ws} window grid")
    n1 = _bn(x, blk.norm1)
    l1 = _bn(conv3x3_32(n1, a.local1[0].weight, B, H, W), a.local1[1])
    l2 = _bn(gemm32(n1, _w2(a.local2[0].weight)), a.local2[1])
    qkv = gemm32(n1, _w2(a.qkv[0].weight))
    table = a.relative_position_bias_table.detach().contiguous()
    if (Hp, Wp) != (H, W):
        qkv = D._reflect_rows(qkv, B, H, W, Hp, Wp)
    o = torch.empty((qkv.shape[0], qkv.shape[1] // 3), dtype=F32, device=x.device)
    call("mfnet_gla_attn_fwd_f32", qkv.data_ptr(), table.data_ptr(), o.data_ptr(), B, Hp, Wp, heads, qkv.shape[1] // (3 * heads), ws,
         float(a.scale), ops._stream(qkv))
    if (Hp, Wp) != (H, W):
        o = o.view(B, Hp, Wp, -1)[:, :H, :W].contiguous().view(B * H * W, -1)
    s = ops.gla_pool_fwd(o, l1, l2, B, H, W)
    k = a.proj[0].weight.shape[-1]
    dwo = ops.dwconv_fwd(s, a.proj[0].weight.detach().reshape(-1, k * k), None, B, H, W, k, True)
    x1 = gemm32(_bn(dwo, a.proj[1]), _w2(a.proj[2].weight), residual=x)
    n2 = _bn(x1, blk.norm2)
    hid = gemm32(n2, _w2(blk.mlp.fc1.weight), bias=blk.mlp.fc1.bias.detach(), act=3)
    return gemm32(hid, _w2(blk.mlp.fc2.weight), bias=blk.mlp.fc2.bias.detach(), residual=x1)


def _fuse32(m, x, res, perm, B, H, W):
    skip = gemm32(res, _w2(m.pre_conv[0].weight))
    f = ops.upfuse_fwd(x, skip, perm, m.weights.detach(), m.eps, B, H, W, out_dtype=F32)
    return _bn(conv3x3_32(f, m.post_conv[0].weight, B, H, W), m.post_conv[1], act=ACT_RELU6)


def _frh32(m, x, res, perm, B, H, W):
    """FeatureRefinementHead (unetformer_mmsam.py:320-354)."""
    P = H * W
    xf = _fuse32(m, x, res, perm, B, H, W)
    sc = _bn(gemm32(xf, _w2(m.shortcut[0].weight)), m.shortcut[1])
    dwo = ops.dwconv_fwd(xf, m.pa[0].weight.detach().reshape(-1, 9), m.pa[0].bias.detach(), B, H, W, 3, False)
    pool, _ = ops.colsum(xf, B, P)
    _, g = ops.se_gate_fwd(pool, P, _w2(m.ca[1][0].weight), None, _w2(m.ca[3][0].weight), None, True)
    pc = ops.frh_gate_fwd(xf, dwo, g, B, P)
    dw2 = ops.dwconv_fwd(pc, m.proj[0].weight.detach().reshape(-1, 9), None, B, H, W, 3, False)
    out = gemm32(_bn(dw2, m.proj[1]), _w2(m.proj[2].weight), residual=sc)
    return ops.act(out, ACT_RELU6)


def decoder32(dec, res1, res2, res3, res4, B, S, h, w, perms=(2, 1)):
    """Decoder.forward (unetformer_mmsam.py:444-457), eval mode (Dropout2d is the identity). S: the token grid (Sh, Sw)."""
    Sh, Sw = (S, S) if isinstance(S, int) else (int(S[0]), int(S[1]))
    if Sh % 2 or Sw % 2 or h != 16 * Sh or w != 16 * Sw:
        raise NotImplementedError("the HIP decoder handles inputs whose height and width are multiples of 32")
    x = _bn(gemm32(res4, _w2(dec.pre_conv[0].weight)), dec.pre_conv[1])
    x = _block32(dec.b4, x, B, Sh // 2, Sw // 2)
    x = _block32(dec.b3, _fuse32(dec.p3, x, res3, 0, B, Sh, Sw), B, Sh, Sw)
    x = _block32(dec.b2, _fuse32(dec.p2, x, res2, perms[1], B, 2 * Sh, 2 * Sw), B, 2 * Sh, 2 * Sw)
    x = _frh32(dec.p1, x, res1, perms[0], B, 4 * Sh, 4 * Sw)
    head = dec.segmentation_head
    hb = _bn(conv3x3_32(x, head[0][0].weight, B, 4 * Sh, 4 * Sw), head[0][1], act=ACT_RELU6)
    wcls = _w2(head[2][0].weight)
    lg = gemm32(hb, wcls)
    return ops.up_final_fwd(lg, B, 4 * Sh, 4 * Sw, wcls.shape[0], 4)


@torch.no_grad()
def forward_fp32(model, x, y):
    """UNetFormer.forward (unetformer_mmsam.py:533-563) in fp32: x (B,3,H,W), y (B,H,W) or (B,3,H,W) -> logits (B,ncls,H,W)."""
    if model.training:
        raise NotImplementedError("forward_fp32 is the eval-mode forward; fp32 training goes through UNetFormer.forward")
    h, w = x.shape[-2:]
    B = x.shape[0]
    deep = encoder32(model.image_encoder, x, y)
    res1, res2, res3, res4 = fpn_fusion32(model, deep)
    return decoder32(model.decoder, res1, res2, res3, res4, B, (deep.shape[1], deep.shape[2]), h, w)
