"""Host side of the multimodal SAM ViT encoder: nn.Modules that mirror the reference's interface
(same class names, constructor arguments, parameter names and two-tensor forwards) and drive the
gfx950 kernels of libmfnet_hip through `ssrs_amd.ops`.

Reference interfaces mirrored (paths relative to /root/reference):
  lora.Linear / lora.MergedLinear      medsam/models/common/loralib/layers.py:93-268
  Adapter                              medsam/models/common/adapter.py:7-25
  LayerNorm2d                          medsam/models/common/layer_norm.py:5-17
  LoraBlock (+Attention, MLPBlock)     medsam/models/ImageEncoder/vit/lora_block.py:10-196
  AdapterFusionBlock                   medsam/models/ImageEncoder/vit/adapter_fusionblock.py:12-198
  ImageEncoderViT / PatchEmbed         medsam/models/sam/modeling/image_encoder.py:21-197

MI355X-first layout: both modalities are stacked into ONE token matrix [2*B*T, D] (they share every
frozen weight), the residual stream stays fp32, GEMM operands are bf16 with fp32 accumulation, and
each residual branch (attention / MLP) is a single autograd node with a hand-written backward, so
no N x N score tensor, padded window or fp32 weight copy ever reaches HBM.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple, Type

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .bank import deliver, derived

BF = torch.bfloat16

# ---------------------------------------------------------------------------------------------
# frozen-weight caches (bf16 copies / transposes, resized tables); rebuilt when a parameter is
# re-loaded (load_state_dict bumps _version) or moved (data_ptr changes)
# ---------------------------------------------------------------------------------------------
def _cached(key, params, build):
    """Derived tensors live in a dict hung on the first source parameter (so they die with it and a
    recycled id() can never alias another model's entry)."""
    store = params[0].__dict__.setdefault("_mfnet_derived", {})
    sig = tuple((p.data_ptr(), p._version, p.device) for p in params)
    hit = store.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1]
    with torch.no_grad():
        val = build()
    store[key] = (sig, val)
    return val


def w_bf16(p: torch.Tensor, tag="") -> torch.Tensor:
    """bf16 copy of a frozen [N,K] weight."""
    return _cached(("w", tag, ops.FP32), (p,), lambda: ops.cast(p.detach().reshape(p.shape[0], -1).contiguous(), BF))


def wT_bf16(p: torch.Tensor, tag="") -> torch.Tensor:
    """bf16 copy of the transpose [K,N] of a frozen [N,K] weight (operand of the dgrad GEMM)."""
    return _cached(("wT", tag, ops.FP32), (p,),
                   lambda: ops.cast(p.detach().reshape(p.shape[0], -1).t().contiguous(), BF))


def _resize_rel(rel: torch.Tensor, need: int) -> torch.Tensor:
    """get_rel_pos's table resize (lora_block.py:260-271): linear interpolation to 2*k-1 rows."""
    if rel.shape[0] == need:
        return rel
    return F.interpolate(rel.t()[None], size=need, mode="linear")[0].t()


def build_rel_table(rel_h: torch.Tensor, rel_w: torch.Tensor, kh: int, kw: int):
    """-> (tbl [NT, hd rounded up to 32] bf16, tblT [hd,NT] bf16, Lh, Lw): rows [0,Lh) the h table, [Lh,Lh+Lw) the w
    table (both resized to 2k-1 rows), zero rows up to a multiple of 32."""
    Lh, Lw = 2 * kh - 1, 2 * kw - 1
    t = torch.cat([_resize_rel(rel_h.detach().float(), Lh), _resize_rel(rel_w.detach().float(), Lw)], 0)
    NT = ((Lh + Lw + 31) // 32) * 32
    hd = t.shape[1]
    hdp = (hd + 31) // 32 * 32                      # the kernels run 32-wide k-steps over the head dim (80 -> 96, zeros)
    full = torch.zeros(NT, hdp, dtype=torch.float32, device=t.device)
    full[:Lh + Lw, :hd] = t
    return full.to(BF).contiguous(), full[:, :hd].t().contiguous().to(BF).contiguous(), Lh, Lw


def _rel_tables(attn: nn.Module, kh: int, kw: int):
    return _cached(("rel", kh, kw), (attn.rel_h, attn.rel_w),
                   lambda: build_rel_table(attn.rel_h, attn.rel_w, kh, kw))


# ---------------------------------------------------------------------------------------------
# forward/backward building blocks (plain functions on tensors; no autograd inside)
# ---------------------------------------------------------------------------------------------
def _ptrs(*ts):
    return tuple(t.data_ptr() for t in ts)


def _stack2(kind, a, b, transpose=False):
    """[2, ...] fp32 stack of two (optionally transposed) 2-D factors, kept per step by the parameter bank."""
    a, b = a.detach(), b.detach()
    if transpose:
        a, b = a.t(), b.t()
    return derived((kind, transpose) + _ptrs(a, b), (2,) + tuple(a.shape), torch.float32,
                   lambda t: [(t[0], a), (t[1], b)], device=a.device)


def _ln_lora_fits(R, C):
    """the fused LayerNorm + down-projection kernel keeps both modalities' A as bf16 in 64 KB of LDS"""
    return 2 * R * C * 2 + 512 <= 65536 and not ops.FP32          # (a bf16 kernel: fp32 training takes the general products)


def _down(x_bf, A_x, A_y, half):
    """a[m] = x[m] @ A_g^T with the modality-specific A (rows [0,half) -> A_x, rest -> A_y)."""
    return ops.skinny_nt(x_bf, _stack2("A", A_x, A_y), rows_per_group=half).contiguous()


def _qkv_lora_V(Bx, By, D, r, transposed=False):
    """Per-modality up-projection of the q,v LoRA as a dense [3D, 2r] factor (zero rows for k; q uses rank
    slots [0,r), v slots [r,2r)), stacked over the two modalities; `transposed` gives [2, 2r, 3D]."""
    Bx, By = Bx.detach(), By.detach()

    def fill(t):
        pairs = []
        for g, Bm in enumerate((Bx, By)):
            if transposed:
                pairs += [(t[g, :r, :D], Bm[:D].t()), (t[g, r:, 2 * D:], Bm[D:].t())]
            else:
                pairs += [(t[g, :D, :r], Bm[:D]), (t[g, 2 * D:, r:], Bm[D:])]
        return pairs
    shape = (2, 2 * r, 3 * D) if transposed else (2, 3 * D, 2 * r)
    return derived(("qkvV", transposed) + _ptrs(Bx, By), shape, torch.float32, fill, device=Bx.device)


def _mix_V(lin, transposed=False):
    """[N, 2r] = [Bx | By] of a lora.Linear (or its transpose [2r, N])."""
    Bx, By = lin.lora_Bx.detach(), lin.lora_By.detach()
    N, r = Bx.shape
    if transposed:
        return derived(("mixVt",) + _ptrs(Bx, By), (2 * r, N), torch.float32,
                       lambda t: [(t[:r], Bx.t()), (t[r:], By.t())], device=Bx.device)
    return derived(("mixV",) + _ptrs(Bx, By), (N, 2 * r), torch.float32,
                   lambda t: [(t[:, :r], Bx), (t[:, r:], By)], device=Bx.device)


def _tn_groups(y_bf, u, half, transposed, scale=1.0):
    o = ops.skinny_tn(y_bf, u, transposed, scale, n_groups=2, lazy=True)     # summed by the gradient delivery
    return o[0], o[1]


def _rel_tables32(attn: nn.Module, kh: int, kw: int):
    """get_rel_pos's resized tables (lora_block.py:260-271) as fp32 [2k-1, hd] matrices; frozen, cached per window size."""
    return _cached(("rel32", kh, kw), (attn.rel_h, attn.rel_w),
                   lambda: (_resize_rel(attn.rel_h.detach().float(), 2 * kh - 1).contiguous(),
                            _resize_rel(attn.rel_w.detach().float(), 2 * kw - 1).contiguous()))


def _attention(qkv, attn_mod, G, Sh, Sw, heads, ws):
    """-> (o, lse, tables). fp32 training (fp32 qkv): the fp32 attention kernel, whose backward recomputes the row statistics
    (lse is None) and takes the two resized fp32 tables instead of the stacked bf16 one."""
    kh, kw = (ws, ws) if ws > 0 else (Sh, Sw)
    hd = qkv.shape[1] // (3 * heads)
    if qkv.dtype == torch.float32:
        th, tw = _rel_tables32(attn_mod, kh, kw)
        return ops.attn_f32(qkv, attn_mod.qkv.bias.detach(), th, tw, G, Sh, Sw, heads, ws, hd ** -0.5), None, (th, tw, 0, 0)
    tbl, tblT, Lh, Lw = _rel_tables(attn_mod, kh, kw)
    o, lse = ops.attn_fwd(qkv, attn_mod.qkv.bias, tbl, G, Sh, Sw, heads, ws, Lh, Lw, hd ** -0.5)
    return o, lse, (tbl, tblT, Lh, Lw)


def _attention_bwd(qkv, attn_mod, tb, o, dO, lse, G, Sh, Sw, heads, ws):
    tbl, tblT, Lh, Lw = tb
    hd = qkv.shape[1] // (3 * heads)
    if qkv.dtype == torch.float32:
        return ops.attn_f32_bwd(qkv, attn_mod.qkv.bias.detach(), tbl, tblT, o, dO, G, Sh, Sw, heads, ws, hd ** -0.5)
    return ops.attn_bwd(qkv, attn_mod.qkv.bias, tbl, tblT, o, dO, lse, G, Sh, Sw, heads, ws, Lh, Lw, hd ** -0.5)


@ops.keeps_precision
class _LoraAttnBranch(torch.autograd.Function):
    """h -> h + proj(attn(qkv_lora(LN1(h))))  for the stacked token matrix (lora_block.py:64-82)."""

    @staticmethod
    def forward(ctx, h, blk, geom, Ax, Ay, Bx, By):
        B, Sh, Sw, heads, ws = geom
        attn = blk.attn
        D = h.shape[-1]
        M = h.numel() // D
        half = M // 2
        r = Ax.shape[0] // 2
        h2 = h.reshape(M, D)
        if 2 * r in (4, 8) and half % 2 == 0 and _ln_lora_fits(2 * r, D):      # LayerNorm and the down-projection a = LN(h) @ A_g^T in one launch
            ln, mean, rstd, a, _ = ops.layernorm_fwd_lora(h2, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, _stack2("A", Ax, Ay))
        else:
            ln, mean, rstd = ops.layernorm_fwd(h2, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
            a = _down(ln, Ax, Ay, half)
        V = _qkv_lora_V(Bx, By, D, r)
        qkv = ops.gemm(ln, w_bf16(attn.qkv.weight), bias=attn.qkv.bias, U=a, V=V, rows_per_group=half,
                       n_groups=2, lr_scale=1.0 / r)
        o, lse, tb = _attention(qkv, attn, 2 * B, Sh, Sw, heads, ws)
        out = ops.gemm(o, w_bf16(attn.proj.weight), bias=attn.proj.bias, residual=h2, out_dtype=torch.float32)
        ctx.save_for_backward(h2, mean, rstd, ln, a, qkv, o, lse)          # (lse: None in fp32 mode)
        ctx.blk, ctx.geom, ctx.tb = blk, geom, tb
        ctx.params = (Ax, Ay, Bx, By)
        return out.view(h.shape)

    @staticmethod
    def backward(ctx, dout):
        h2, mean, rstd, ln, a, qkv, o, lse = ctx.saved_tensors
        blk, (B, Sh, Sw, heads, ws) = ctx.blk, ctx.geom
        Ax, Ay, Bx, By = ctx.params
        attn = blk.attn
        M, D = h2.shape
        half = M // 2
        r = Ax.shape[0] // 2
        hd = D // heads
        d2 = dout.reshape(M, D).contiguous()
        d_bf = ops.bf16_of(d2)
        dO = ops.gemm(d_bf, wT_bf16(attn.proj.weight))
        dqkv = _attention_bwd(qkv, attn, ctx.tb, o, dO, lse, 2 * B, Sh, Sw, heads, ws)
        s = 1.0 / r
        if 2 * r == 8 and not ops.FP32:
            # one pass over dqkv for d(xA^T) = dqkv . B (per modality) and dB = dqkv^T . a; one over ln for dA^T = ln^T . da
            dV, pp = ops.lora_side(dqkv, U=a, Vt=_qkv_lora_V(Bx, By, D, r, transposed=True), n_groups=2, scale=s)
            da = ops.reduce_parts(pp, s)
            dVx, dVy = dV[0], dV[1]
            dA = ops.lora_side(ln, U=da, n_groups=2)[0]
            dAx, dAy = dA[0].t(), dA[1].t()
        else:
            da = ops.skinny_nt(dqkv, _qkv_lora_V(Bx, By, D, r, transposed=True), scale=s, rows_per_group=half).contiguous()
            dVx, dVy = _tn_groups(dqkv, a, half, False, s)
            dAx, dAy = _tn_groups(ln, da, half, True)
        rows_q, rows_v = slice(0, D), slice(D, 2 * D)
        grads = deliver((Ax, Ay, Bx, By), (dAx, dAy, [(rows_q, dVx[:D, :r]), (rows_v, dVx[2 * D:, r:])],
                                           [(rows_q, dVy[:D, :r]), (rows_v, dVy[2 * D:, r:])]))
        if not ctx.needs_input_grad[0]:          # first block over a frozen patch embedding: nothing upstream wants dh
            return (None, None, None) + grads
        dln = ops.gemm(dqkv, wT_bf16(attn.qkv.weight), U=da, V=_stack2("A", Ax, Ay, transpose=True), rows_per_group=half, n_groups=2)
        dh, dh_bf = ops.layernorm_bwd(dln, h2, mean, rstd, blk.norm1.weight, dres=d2, want_bf16=True)
        ops.remember_bf16_twin(dh, dh_bf)
        return (dh.view(dout.shape), None, None) + grads


def _mix_linear_fwd(x_bf, lin, half, act, residual, aU=None):
    """lora.Linear forward on stacked rows: returns (out, pre|None, saved). aU: (a, U) if the producer of x_bf already made
    them (ops.layernorm_fwd_lora)."""
    r = lin.lora_Ax.shape[0]
    K = x_bf.shape[-1]
    if aU is not None:
        a, U = aU
    elif not ops.FP32 and r in (4, 8) and half % 2 == 0 and K % 8 == 0 and K <= 4096 and 2 * r * K * 2 + 512 <= 96 * 1024 and x_bf.shape[0] == 2 * half:
        a, U = ops.lora_down_mix(x_bf, _stack2("A", lin.lora_Ax, lin.lora_Ay), lin.lora_wx.detach(), lin.lora_wy.detach())
    else:
        a = _down(x_bf, lin.lora_Ax, lin.lora_Ay, half)
        U = ops.lora_mix_fwd(a, lin.lora_wx, lin.lora_wy)
    V = _mix_V(lin)                                                # [N, 2r]
    if act:
        out, pre = ops.gemm(x_bf, w_bf16(lin.weight), bias=lin.bias, U=U, V=V, act=1, want_pre=True)
    else:
        out = ops.gemm(x_bf, w_bf16(lin.weight), bias=lin.bias, U=U, V=V, residual=residual,
                       out_dtype=torch.float32 if residual is not None else BF)
        pre = None
    return out, pre, (a, U)


def _mix_linear_bwd(d_bf, x_bf, lin, saved, half, gelu_aux=None):
    """-> (dx bf16, (dAx, dAy, dBx, dBy, dwx, dwy) as strided views). d_bf: grad of the layer output (bf16,
    after any activation backward of the NEXT layer); gelu_aux: this layer's INPUT pre-activation if the
    input was GELU(pre) (then dx is the grad w.r.t. pre)."""
    a, U = saved
    r = a.shape[1]
    if 2 * r == 8 and not ops.FP32:
        dV, pp = ops.lora_side(d_bf, U=U, Vt=_mix_V(lin, transposed=True))      # d_bf is read once for P = d . [Bx|By] and dB
        da, dw = ops.lora_mix_bwd(pp, a, lin.lora_wx, lin.lora_wy)
        dA = ops.lora_side(x_bf, U=da, n_groups=2)[0]
        dAx, dAy = dA[0].t(), dA[1].t()
    else:
        P = ops.skinny_nt(d_bf, _mix_V(lin, transposed=True)).contiguous()
        da, dw = ops.lora_mix_bwd(P, a, lin.lora_wx, lin.lora_wy)
        dV = ops.skinny_tn(d_bf, U, False, lazy=True)
        dAx, dAy = _tn_groups(x_bf, da, half, True)
    dx = ops.gemm(d_bf, wT_bf16(lin.weight), U=da, V=_stack2("A", lin.lora_Ax, lin.lora_Ay, transpose=True),
                  rows_per_group=half, n_groups=2, act=2 if gelu_aux is not None else 0, aux=gelu_aux)
    return dx, (dAx, dAy, dV[:, :r], dV[:, r:], dw[0:1], dw[1:2])


@ops.keeps_precision
class _LoraMlpBranch(torch.autograd.Function):
    """h -> h + lin2(gelu(lin1(LN2(h))))  with the cross-modal LoRA mix (lora_block.py:83-85,104-108)."""

    @staticmethod
    def forward(ctx, h, blk, *params):
        D = h.shape[-1]
        M = h.numel() // D
        half = M // 2
        h2 = h.reshape(M, D)
        l1 = blk.mlp.lin1
        if l1.lora_Ax.shape[0] in (4, 8) and half % 2 == 0 and _ln_lora_fits(l1.lora_Ax.shape[0], D):     # LayerNorm, lin1's down-projection and its cross-modal mix in one launch
            ln, mean, rstd, a1, U1 = ops.layernorm_fwd_lora(h2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps,
                                                            _stack2("A", l1.lora_Ax, l1.lora_Ay), l1.lora_wx.detach(), l1.lora_wy.detach())
            aU = (a1, U1)
        else:
            ln, mean, rstd = ops.layernorm_fwd(h2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            aU = None
        act, pre, s1 = _mix_linear_fwd(ln, l1, half, True, None, aU)
        out, _, s2 = _mix_linear_fwd(act, blk.mlp.lin2, half, False, h2)
        ctx.save_for_backward(h2, mean, rstd, ln, act, pre, *s1, *s2)
        ctx.blk = blk
        return out.view(h.shape)

    @staticmethod
    def backward(ctx, dout):
        h2, mean, rstd, ln, act, pre, a1, U1, a2, U2 = ctx.saved_tensors
        blk = ctx.blk
        M, D = h2.shape
        half = M // 2
        d2 = dout.reshape(M, D).contiguous()
        d_bf = ops.bf16_of(d2)
        dpre, g2 = _mix_linear_bwd(d_bf, act, blk.mlp.lin2, (a2, U2), half, gelu_aux=pre)
        dln, g1 = _mix_linear_bwd(dpre, ln, blk.mlp.lin1, (a1, U1), half)
        dh, dh_bf = ops.layernorm_bwd(dln, h2, mean, rstd, blk.norm2.weight, dres=d2, want_bf16=True)
        ops.remember_bf16_twin(dh, dh_bf)
        return (dh.view(dout.shape), None) + deliver(blk._mlp_params(), g1 + g2)     # order of LoraBlock._mlp_params()


@ops.keeps_precision
class _Neck(torch.autograd.Function):
    """frozen neck: 1x1 conv -> LayerNorm2d -> 3x3 conv -> LayerNorm2d on channels-last tokens
    (image_encoder.py:104-120,142-143). h [G,Sh,Sw,D] fp32 -> [G,Sh,Sw,256] fp32."""

    @staticmethod
    def forward(ctx, h, enc):
        G, Sh, Sw, D = h.shape
        M = G * Sh * Sw
        neck = enc.neck
        C = neck[0].weight.shape[0]
        hb = ops.cast(h.reshape(M, D).contiguous(), BF)
        n1 = ops.gemm(hb, w_bf16(neck[0].weight), out_dtype=torch.float32)
        l1, m1, r1 = ops.layernorm_fwd(n1, neck[1].weight, neck[1].bias, neck[1].eps)
        if C % 64 == 0:                  # implicit-GEMM 3x3 conv: no column tensor
            n2 = ops.conv3x3(l1, _neck3x3_w(neck[2].weight), G, Sh, Sw)
        else:
            n2 = ops.gemm(ops.im2col3x3(l1, G, Sh, Sw), _neck3x3_w(neck[2].weight), out_dtype=torch.float32)
        out, m3, r3 = ops.layernorm_fwd(n2, neck[3].weight, neck[3].bias, neck[3].eps, out_dtype=torch.float32)
        ctx.save_for_backward(n1, m1, r1, n2, m3, r3)
        ctx.enc, ctx.shape = enc, (G, Sh, Sw, D, C)
        return out.view(G, Sh, Sw, C)

    @staticmethod
    def backward(ctx, dout):
        n1, m1, r1, n2, m3, r3 = ctx.saved_tensors
        G, Sh, Sw, D, C = ctx.shape
        neck = ctx.enc.neck
        M = G * Sh * Sw
        _, dn2 = ops.layernorm_bwd(dout.reshape(M, C).contiguous(), n2, m3, r3, neck[3].weight, want_f32=False, want_bf16=True)
        if C % 64 == 0:
            dl1 = ops.conv3x3(dn2, _neck3x3_wflipT(neck[2].weight), G, Sh, Sw)
        else:
            dl1 = ops.col2im3x3(ops.gemm(dn2, _neck3x3_wT(neck[2].weight)), G, Sh, Sw, C)
        _, dn1 = ops.layernorm_bwd(dl1, n1, m1, r1, neck[1].weight, want_f32=False, want_bf16=True)
        dh = ops.gemm(dn1, wT_bf16(neck[0].weight), out_dtype=torch.float32)
        return dh.view(G, Sh, Sw, D), None


def _neck3x3_w(w):     # [O, I, 3, 3] -> [O, 9*I] with k = tap*I + c  (im2col3x3 order)
    return _cached(("n3", ops.FP32), (w,), lambda: ops.cast(w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous(), BF))


def _neck3x3_wflipT(w):   # [O, I, 3, 3] -> [I, 9*O] with k = (8 - tap)*O + o: the data gradient as a 3x3 conv of dy
    return _cached(("n3fT", ops.FP32), (w,), lambda: ops.cast(w.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous(), BF))


def _neck3x3_wT(w):
    return _cached(("n3T", ops.FP32), (w,), lambda: ops.cast(w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).t().contiguous(), BF))


# ---------------------------------------------------------------------------------------------
# loralib layers (two-tensor forward, loralib/layers.py)
# ---------------------------------------------------------------------------------------------
class LoRALayer:
    def __init__(self, r: int, lora_alpha: int, lora_dropout: float, merge_weights: bool):
        if lora_dropout and lora_dropout > 0.0:
            raise NotImplementedError("lora_dropout > 0 is never used by MFNet and is not implemented")
        self.r = r
        self.lora_alpha = lora_alpha
        self.lora_dropout = lambda x: x
        self.merged = False
        self.merge_weights = merge_weights


def _check_rank(r):
    """`-mid_dim` (lora_block.py:44-47, medsam/cfg.py:47): any rank from 1 to 16. Ranks 4 and 8 take the fused LayerNorm /
    row-streaming side kernels, the others the general skinny products (the GEMM epilogue adds up to 32 stacked rank columns, 8 per
    extra k-step; factors whose width is not a multiple of 4 are zero-padded on the host side)."""
    if not 1 <= int(r) <= 16:
        raise ValueError(f"LoRA rank {r} unsupported by the HIP path (1 .. 16; default 4)")


class Linear(nn.Linear, LoRALayer):
    """lora.Linear (loralib/layers.py:93-167): dense layer shared by both modalities plus a
    per-modality rank-r update, cross-mixed by the trainable scalars lora_wx / lora_wy."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0, fan_in_fan_out=False,
                 merge_weights=True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        LoRALayer.__init__(self, r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout, merge_weights=merge_weights)
        if fan_in_fan_out:
            raise NotImplementedError("fan_in_fan_out is unused by MFNet")
        self.fan_in_fan_out = False
        if r > 0:
            _check_rank(r)
            self.lora_Ax = nn.Parameter(self.weight.new_zeros((r, in_features)))
            self.lora_Ay = nn.Parameter(self.weight.new_zeros((r, in_features)))
            self.lora_Bx = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.lora_By = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
            for a in (self.lora_Ax, self.lora_Ay):
                nn.init.kaiming_uniform_(a, a=math.sqrt(5))
        self.lora_wx = nn.Parameter(torch.full((1,), 0.5))
        self.lora_wy = nn.Parameter(torch.full((1,), 0.5))

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        h = torch.stack([x, y])
        out = _StandaloneMixLinear.apply(h, self, self.lora_Ax, self.lora_Ay, self.lora_Bx, self.lora_By,
                                         self.lora_wx, self.lora_wy)
        return out[0], out[1]


@ops.keeps_precision
class _StandaloneMixLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, lin, *params):
        K = h.shape[-1]
        M = h.numel() // K
        xb = ops.cast(h.reshape(M, K).contiguous().float(), BF)
        out, _, saved = _mix_linear_fwd(xb, lin, M // 2, False, None)
        ctx.save_for_backward(xb, *saved)
        ctx.lin = lin
        return out.float().view(*h.shape[:-1], -1)

    @staticmethod
    def backward(ctx, dout):
        xb, a, U = ctx.saved_tensors
        N = dout.shape[-1]
        M = dout.numel() // N
        lin = ctx.lin
        d_bf = ops.cast(dout.reshape(M, N).contiguous(), BF)
        dx, g = _mix_linear_bwd(d_bf, xb, lin, (a, U), M // 2)
        params = (lin.lora_Ax, lin.lora_Ay, lin.lora_Bx, lin.lora_By, lin.lora_wx, lin.lora_wy)
        return (dx.float().view(*dout.shape[:-1], -1), None) + deliver(params, g)


class MergedLinear(nn.Linear, LoRALayer):
    """lora.MergedLinear (loralib/layers.py:169-268) with enable_lora=[True, False, True]."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0, enable_lora=(False,),
                 fan_in_fan_out=False, merge_weights=True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        LoRALayer.__init__(self, r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout, merge_weights=merge_weights)
        enable_lora = list(enable_lora)
        assert out_features % len(enable_lora) == 0, "The length of enable_lora must divide out_features"
        if fan_in_fan_out:
            raise NotImplementedError("fan_in_fan_out is unused by MFNet")
        if r > 0 and enable_lora != [True, False, True]:
            raise NotImplementedError("the HIP path implements enable_lora=[True, False, True] (q and v), as MFNet uses")
        self.enable_lora = enable_lora
        self.fan_in_fan_out = False
        if r > 0:
            _check_rank(r)
            n = sum(enable_lora)
            per = out_features // len(enable_lora)
            self.lora_Ax = nn.Parameter(self.weight.new_zeros((r * n, in_features)))
            self.lora_Bx = nn.Parameter(self.weight.new_zeros((per * n, r)))
            self.lora_Ay = nn.Parameter(self.weight.new_zeros((r * n, in_features)))
            self.lora_By = nn.Parameter(self.weight.new_zeros((per * n, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
            ind = self.weight.new_zeros((out_features,), dtype=torch.bool).view(len(enable_lora), -1)
            ind[enable_lora, :] = True
            self.lora_ind = ind.view(-1)
            for a in (self.lora_Ax, self.lora_Ay):
                nn.init.kaiming_uniform_(a, a=math.sqrt(5))

    @torch.no_grad()
    def forward(self, x: torch.Tensor, y: torch.Tensor):
        """layers.py:244-268, forward only (training runs through _LoraAttnBranch, which owns the backward): both
        modalities in one GEMM launch whose epilogue adds the per-modality rank-r update of the q and v thirds."""
        h = torch.stack([x, y])
        K = h.shape[-1]
        M = h.numel() // K
        xb = ops.cast(h.reshape(M, K).contiguous().float(), BF)
        if self.r > 0:
            a = _down(xb, self.lora_Ax, self.lora_Ay, M // 2)
            V = _qkv_lora_V(self.lora_Bx, self.lora_By, self.out_features // 3, self.r)
            out = ops.gemm(xb, w_bf16(self.weight), bias=self.bias, U=a, V=V, rows_per_group=M // 2, n_groups=2,
                           lr_scale=self.scaling, out_dtype=torch.float32)
        else:
            out = ops.gemm(xb, w_bf16(self.weight), bias=self.bias, out_dtype=torch.float32)
        out = out.view(*h.shape[:-1], -1)
        return out[0], out[1]


class Adapter(nn.Module):
    """common/adapter.py:7-25."""

    def __init__(self, D_features, mlp_ratio=0.25, act_layer=nn.GELU, skip_connect=True):
        super().__init__()
        self.skip_connect = skip_connect
        D_hidden = int(D_features * mlp_ratio)
        self.act = act_layer()
        self.D_fc1 = nn.Linear(D_features, D_hidden)
        self.D_fc2 = nn.Linear(D_hidden, D_features)

    @torch.no_grad()
    def forward(self, x):
        """adapter.py:16-25, forward only (AdapterFusionBlock's fused branches own the backward): fc2(GELU(fc1(x))) (+ x)."""
        if not isinstance(self.act, nn.GELU):
            raise NotImplementedError("the HIP epilogue implements the erf GELU MFNet uses")
        K = x.shape[-1]
        x2 = x.reshape(-1, K).contiguous().float()
        hid = ops.gemm(ops.cast(x2, BF), w_bf16(self.D_fc1.weight), bias=self.D_fc1.bias, act=1)
        out = ops.gemm(hid, w_bf16(self.D_fc2.weight), bias=self.D_fc2.bias, residual=x2 if self.skip_connect else None,
                       out_dtype=torch.float32)
        return out.view(x.shape)


class LayerNorm2d(nn.Module):
    """common/layer_norm.py:5-17 (NCHW interface; computed channels-last by the LayerNorm kernel)."""

    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .decoder_ops import layer_norm_nhwc
        return layer_norm_nhwc(x.permute(0, 2, 3, 1).contiguous(), self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------------------------
@ops.keeps_precision
class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.gelu_bwd(ctx.saved_tensors[0], dy.contiguous().float())


_gelu = _Gelu.apply


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim: int, mlp_dim: int, act: Type[nn.Module] = nn.GELU, lora_rank: int = 4) -> None:
        super().__init__()
        self.lin1 = Linear(embedding_dim, mlp_dim, r=lora_rank)
        self.lin2 = Linear(mlp_dim, embedding_dim, r=lora_rank)
        self.act = act()

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        """lora_block.py:104-108 (standalone; the block fuses it into _LoraMlpBranch)."""
        x, y = self.lin1(x, y)
        x, y = _gelu(x), _gelu(y)
        return self.lin2(x, y)


class Attention(nn.Module):
    """LoRA attention (lora_block.py:111-196): parameter container; compute is in _LoraAttnBranch."""

    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True,
                 lora_rank=4, input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = MergedLinear(dim, dim * 3, bias=qkv_bias, r=lora_rank, enable_lora=[True, False, True])
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if self.use_rel_pos:
            assert input_size is not None, "Input size must be provided if using relative positional encoding."
            self.rel_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))
        else:
            raise NotImplementedError("MFNet always builds SAM with use_rel_pos=True (build_sam.py:84)")


def _geom(h, heads, ws):
    _, B, Sh, Sw, _ = h.shape
    return (B, Sh, Sw, heads, ws)


class LoraBlock(nn.Module):
    """lora_block.py:10-87. forward(x, y) keeps the reference's two-tensor interface; the encoder
    calls forward_stacked on the [2,B,H,W,C] residual stream directly."""

    def __init__(self, args, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm,
                 act_layer=nn.GELU, use_rel_pos=False, rel_pos_zero_init=True, window_size=0,
                 input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.norm1 = norm_layer(dim)
        lora_rank = args.mid_dim if args.mid_dim is not None else 4
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init, lora_rank=lora_rank,
                              input_size=(64, 64) if window_size == 0 else (window_size, window_size))
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer, lora_rank=lora_rank)
        self.window_size = window_size
        self.num_heads = num_heads

    def _mlp_params(self):
        l1, l2 = self.mlp.lin1, self.mlp.lin2
        return (l1.lora_Ax, l1.lora_Ay, l1.lora_Bx, l1.lora_By, l1.lora_wx, l1.lora_wy,
                l2.lora_Ax, l2.lora_Ay, l2.lora_Bx, l2.lora_By, l2.lora_wx, l2.lora_wy)

    def forward_stacked(self, h: torch.Tensor) -> torch.Tensor:
        q = self.attn.qkv
        h = _LoraAttnBranch.apply(h, self, _geom(h, self.num_heads, self.window_size),
                                  q.lora_Ax, q.lora_Ay, q.lora_Bx, q.lora_By)
        return _LoraMlpBranch.apply(h, self, *self._mlp_params())

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        h = self.forward_stacked(torch.stack([x, y]).float().contiguous())
        return h[0], h[1]


class PatchEmbed(nn.Module):
    """image_encoder.py:166-197 (parameter container; the im2col GEMM runs in ImageEncoderViT)."""

    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768) -> None:
        super().__init__()
        assert tuple(kernel_size) == tuple(stride) and tuple(padding) == (0, 0) and in_chans == 3
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


class ImageEncoderViT(nn.Module):
    """image_encoder.py:21-144: forward(x, y) -> (fx, fy), each (B, out_chans, H/16, W/16)."""

    def __init__(self, args, img_size=1024, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, out_chans=256, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                 use_abs_pos=True, use_rel_pos=False, rel_pos_zero_init=True, window_size=0,
                 global_attn_indexes=()) -> None:
        super().__init__()
        self.img_size = img_size
        self.args = args
        self.embed_dim = embed_dim
        self.patch_size = patch_size
        self.patch_embed = PatchEmbed(kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size),
                                      in_chans=in_chans, embed_dim=embed_dim)
        self.pos_embed = None
        if use_abs_pos:
            self.pos_embed = nn.Parameter(torch.zeros(1, 1024 // patch_size, 1024 // patch_size, embed_dim))
        if args.mod == "sam_adpt":
            from .vit_adapter import AdapterFusionBlock
            block_class = AdapterFusionBlock
        elif args.mod == "sam_lora":
            block_class = LoraBlock
        else:
            raise NotImplementedError(f"mod={args.mod!r}: the HIP path implements 'sam_lora' (MMLoRA) and "
                                      "'sam_adpt' (MMAdapter), the two dual-modality variants MFNet can run")
        self.blocks = nn.ModuleList()
        for i in range(depth):
            self.blocks.append(block_class(
                args=args, dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                norm_layer=norm_layer, act_layer=act_layer, use_rel_pos=use_rel_pos,
                rel_pos_zero_init=rel_pos_zero_init,
                window_size=window_size if i not in global_attn_indexes else 0,
                input_size=(img_size // patch_size, img_size // patch_size)))
        self.neck = nn.Sequential(
            nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False), LayerNorm2d(out_chans),
            nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False), LayerNorm2d(out_chans))

    def _pos(self, Sh, Sw):
        """abs-pos-embed bicubically resized to the token grid (image_encoder.py:126-135); frozen, so
        computed once per grid size and cached as a [T, D] fp32 matrix."""
        p = self.pos_embed
        return _cached(("pos", Sh, Sw), (p,), lambda: F.interpolate(
            p.detach().permute(0, 3, 1, 2), size=(Sh, Sw), mode="bicubic", align_corners=False
        ).permute(0, 2, 3, 1).reshape(Sh * Sw, -1).contiguous())

    def embed(self, rgb: torch.Tensor, dsm: torch.Tensor) -> torch.Tensor:
        """patch-embed both modalities (a (B,H,W) DSM plane is replicated to 3 channels by the im2col; a (B,3,H,W) second
        modality is read as it is) + abs-pos add
        -> stacked residual stream [2, B, Sh, Sw, D] fp32."""
        B, _, H, W = rgb.shape
        P = self.patch_size
        Sh, Sw = H // P, W // P
        with torch.no_grad():
            cols = ops.im2col_patch(rgb.float().contiguous(), dsm.float().contiguous(), P)
            pe = self.patch_embed.proj
            if self.pos_embed is not None:
                h = ops.gemm(cols, w_bf16(pe.weight), bias=pe.bias, residual=self._pos(Sh, Sw), res_row_mod=Sh * Sw,
                             out_dtype=torch.float32)
            else:
                h = ops.gemm(cols, w_bf16(pe.weight), bias=pe.bias, out_dtype=torch.float32)
        return h.view(2, B, Sh, Sw, -1)

    def forward_stacked(self, rgb, dsm):
        h = self.embed(rgb, dsm)
        for blk in self.blocks:
            h = blk.forward_stacked(h)
        two, B, Sh, Sw, D = h.shape
        return _Neck.apply(h.view(2 * B, Sh, Sw, D), self)          # [2B, Sh, Sw, C] channels-last

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        """x, y: (B,3,H,W) (image_encoder.py:122-144). MFNet feeds y = the DSM repeated to 3 channels
        (unetformer_mmsam.py:535) and calls forward_stacked with the plane itself; here y is taken as the 3-channel image
        the interface promises - the patch gather reads its own channels, no host-side check of the data."""
        B = x.shape[0]
        out = self.forward_stacked(x, y)
        fx, fy = out[:B], out[B:]
        return fx.permute(0, 3, 1, 2), fy.permute(0, 3, 1, 2)
