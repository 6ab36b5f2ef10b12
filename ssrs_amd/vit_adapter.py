"""MMAdapter block drop-in (reference: medsam/models/ImageEncoder/vit/adapter_fusionblock.py:12-198).

Shared frozen attention per modality, Img_/DSM_Adapter (skip) on the attention output, and the
MLPx_/MLPy_Adapter pair cross-mixed by the scalars wx_/wy_Adapter next to the frozen MLP.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .bank import deliver, derived, weight_bf16
from .vit import BF, Adapter, _attention, _attention_bwd, _geom, w_bf16, wT_bf16


class Attention(nn.Module):
    """adapter_fusionblock.py:142-198: plain (frozen) qkv / proj with decomposed rel-pos."""

    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True,
                 input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if not use_rel_pos:
            raise NotImplementedError("MFNet always builds SAM with use_rel_pos=True (build_sam.py:84)")
        assert input_size is not None, "Input size must be provided if using relative positional encoding."
        self.rel_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
        self.rel_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim: int, mlp_dim: int, act=nn.GELU) -> None:
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)
        self.act = act()

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """adapter_fusionblock.py:128-139, forward only (frozen weights; the block fuses it into _AdapterMlpBranch)."""
        K = x.shape[-1]
        xb = ops.cast(x.reshape(-1, K).contiguous().float(), BF)
        hid = ops.gemm(xb, w_bf16(self.lin1.weight), bias=self.lin1.bias, act=1)
        return ops.gemm(hid, w_bf16(self.lin2.weight), bias=self.lin2.bias, out_dtype=torch.float32).view(*x.shape[:-1], -1)


# ---- trainable per-modality dense layer pair (x rows use (Wx, bx), y rows (Wy, by)): forward + (dgrad, wgrad, bias grad)
def _w2(Wx, Wy, transpose=False):
    """bf16 [2, N, K] GEMM image of the two modalities' weights (or of their transposes), kept per step by the bank."""
    a, b = Wx.detach(), Wy.detach()
    if transpose:
        a, b = a.t(), b.t()
    return derived(("w2", transpose, a.data_ptr(), b.data_ptr()), (2,) + tuple(a.shape), BF, lambda t: [(t[0], a), (t[1], b)], device=a.device)


def _b2(bx, by):
    a, b = bx.detach(), by.detach()
    return derived(("b2", a.data_ptr(), b.data_ptr()), (2, a.shape[0]), torch.float32, lambda t: [(t[0], a), (t[1], b)], device=a.device)


def _dense_fwd(x_bf, Wx, bx, Wy, by, half, act=False, residual=None, out=None, out_dtype=BF):
    """one grouped launch for both modalities (adapter.py:16-25 with Img_ / DSM_ or MLPx_ / MLPy_ weights)"""
    if act:
        return ops.gemm_grouped(x_bf, _w2(Wx, Wy), bias=_b2(bx, by), rows_per_group=half, act=1, want_pre=True)
    return ops.gemm_grouped(x_bf, _w2(Wx, Wy), bias=_b2(bx, by), rows_per_group=half, residual=residual, out=out, out_dtype=out_dtype)


def _dense_bwd(d_bf, x_bf, Wx, Wy, half, gelu_aux=None):
    """-> (dx bf16 for both halves, (dWx, dWy) fp32 split sums, (dbx, dby) fp32) for y = x W_g^T + b_g."""
    dx = ops.gemm_grouped(d_bf, _w2(Wx, Wy, transpose=True), rows_per_group=half, act=2 if gelu_aux is not None else 0, aux=gelu_aux)
    # both modalities' products in one launch, the bias gradients (column sums of d) from the same fragments; the partial
    # tiles / rows of the row splits are summed by the delivery
    dW, db = ops.gemm_tn_grouped(d_bf, x_bf, 2, want_bias=True)
    return dx, dW, db


@ops.keeps_precision
class _AdapterAttnBranch(torch.autograd.Function):
    """h -> h + A(o), o = proj(attn(qkv(LN1(h)))), A(o) = o + fc2(gelu(fc1(o))) with Img_Adapter weights
    on the RGB rows and DSM_Adapter weights on the DSM rows (adapter_fusionblock.py:78-100)."""

    @staticmethod
    def forward(ctx, h, blk, geom, w1x, b1x, w2x, b2x, w1y, b1y, w2y, b2y):
        B, Sh, Sw, heads, ws = geom
        attn = blk.attn
        D = h.shape[-1]
        M = h.numel() // D
        half = M // 2
        h2 = h.reshape(M, D)
        ln, mean, rstd = ops.layernorm_fwd(h2, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        qkv = ops.gemm(ln, w_bf16(attn.qkv.weight), bias=attn.qkv.bias)
        o, lse, tb = _attention(qkv, attn, 2 * B, Sh, Sw, heads, ws)
        # proj: bf16 copy of o_proj (adapter input) + fp32 h + o_proj
        hm = torch.empty_like(h2)
        _, op = ops.gemm(o, w_bf16(attn.proj.weight), bias=attn.proj.bias, residual=h2, out=hm, want_pre=True)
        a, p = _dense_fwd(op, w1x, b1x, w1y, b1y, half, act=True)
        _dense_fwd(a, w2x, b2x, w2y, b2y, half, residual=hm, out=hm)
        ctx.save_for_backward(h2, mean, rstd, qkv, o, lse, op, a, p, w1x, w2x, w1y, w2y)
        ctx.blk, ctx.geom, ctx.tb = blk, geom, tb
        ctx.params = (w1x, b1x, w2x, b2x, w1y, b1y, w2y, b2y)
        return hm.view(h.shape)

    @staticmethod
    def backward(ctx, dout):
        (h2, mean, rstd, qkv, o, lse, op, a, p, w1x, w2x, w1y, w2y) = ctx.saved_tensors
        blk, (B, Sh, Sw, heads, ws) = ctx.blk, ctx.geom
        attn = blk.attn
        M, D = h2.shape
        half = M // 2
        hd = D // heads
        d2 = dout.reshape(M, D).contiguous()
        d_bf = ops.bf16_of(d2)
        dop = torch.empty_like(d_bf)
        dpre, dW2, db2 = _dense_bwd(d_bf, a, w2x, w2y, half, gelu_aux=p)
        dxo, dW1, db1 = _dense_bwd(dpre, op, w1x, w1y, half)
        ops.add_bf16(dxo, d_bf, dop)                           # adapter skip: d o_proj = d + fc1-path
        grads = [dW1[0], db1[0], dW2[0], db2[0], dW1[1], db1[1], dW2[1], db2[1]]
        if not ctx.needs_input_grad[0]:          # first block over a frozen patch embedding: the (frozen) attention needs no backward
            return (None, None, None) + deliver(ctx.params, grads)
        dO = ops.gemm(dop, wT_bf16(attn.proj.weight))
        dqkv = _attention_bwd(qkv, attn, ctx.tb, o, dO, lse, 2 * B, Sh, Sw, heads, ws)
        dln = ops.gemm(dqkv, wT_bf16(attn.qkv.weight))
        dh, dh_bf = ops.layernorm_bwd(dln, h2, mean, rstd, blk.norm1.weight, dres=d2, want_bf16=True)
        ops.remember_bf16_twin(dh, dh_bf)
        return (dh.view(dout.shape), None, None) + deliver(ctx.params, grads)


@ops.keeps_precision
class _AdapterMlpBranch(torch.autograd.Function):
    """h -> h + mlp(LN2(h)) + mix(MLPx_Adapter(xn), MLPy_Adapter(yn))  (adapter_fusionblock.py:101-108)."""

    @staticmethod
    def forward(ctx, h, blk, w1x, b1x, w2x, b2x, w1y, b1y, w2y, b2y, wx, wy):
        D = h.shape[-1]
        M = h.numel() // D
        half = M // 2
        h2 = h.reshape(M, D)
        ln, mean, rstd = ops.layernorm_fwd(h2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        mlp = blk.mlp
        act, pre = ops.gemm(ln, w_bf16(mlp.lin1.weight), bias=mlp.lin1.bias, act=1, want_pre=True)
        hm = ops.gemm(act, w_bf16(mlp.lin2.weight), bias=mlp.lin2.bias, residual=h2, out_dtype=torch.float32)
        ada = torch.empty((M, D), dtype=torch.float32, device=h.device)
        a, p = _dense_fwd(ln, w1x, b1x, w1y, b1y, half, act=True)
        _dense_fwd(a, w2x, b2x, w2y, b2y, half, out=ada)
        out = ops.adapter_mix(hm, ada, wx, wy)
        ctx.save_for_backward(h2, mean, rstd, ln, pre, ada, a, p, w1x, w2x, w1y, w2y, wx, wy)
        ctx.blk = blk
        ctx.params = (w1x, b1x, w2x, b2x, w1y, b1y, w2y, b2y, wx, wy)
        return out.view(h.shape)

    @staticmethod
    def backward(ctx, dout):
        (h2, mean, rstd, ln, pre, ada, a, p, w1x, w2x, w1y, w2y, wx, wy) = ctx.saved_tensors
        blk = ctx.blk
        mlp = blk.mlp
        M, D = h2.shape
        half = M // 2
        d2 = dout.reshape(M, D).contiguous()
        d_bf = ops.bf16_of(d2)
        dada, dw = ops.adapter_mix_bwd(d2, ada, wx, wy)
        # frozen MLP: dgrad only
        dpre = ops.gemm(d_bf, wT_bf16(mlp.lin2.weight), act=2, aux=pre)
        dln = ops.gemm(dpre, wT_bf16(mlp.lin1.weight))
        dp, dW2, db2 = _dense_bwd(dada, a, w2x, w2y, half, gelu_aux=p)
        dxl, dW1, db1 = _dense_bwd(dp, ln, w1x, w1y, half)
        ops.add_bf16(dxl, dln, dln)
        grads = [dW1[0], db1[0], dW2[0], db2[0], dW1[1], db1[1], dW2[1], db2[1]]
        dh, dh_bf = ops.layernorm_bwd(dln, h2, mean, rstd, blk.norm2.weight, dres=d2, want_bf16=True)
        ops.remember_bf16_twin(dh, dh_bf)
        return (dh.view(dout.shape), None) + deliver(ctx.params, grads + [dw[0:1], dw[1:2]])


def _ad_params(a: Adapter):
    return (a.D_fc1.weight, a.D_fc1.bias, a.D_fc2.weight, a.D_fc2.bias)


class AdapterFusionBlock(nn.Module):
    """adapter_fusionblock.py:12-114; forward(x, y) -> (x, y) on (B,H,W,C) tensors."""

    def __init__(self, args, dim, num_heads, mlp_ratio=4.0, scale=0.5, qkv_bias=True, norm_layer=nn.LayerNorm,
                 act_layer=nn.GELU, use_rel_pos=False, rel_pos_zero_init=True, window_size=0,
                 input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.args = args
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        adapter_dim = args.mid_dim if args.mid_dim is not None else dim
        if adapter_dim != dim:
            raise NotImplementedError("mid_dim != embed dim makes the reference's adapters shape-inconsistent")
        self.MLPx_Adapter = Adapter(adapter_dim, skip_connect=False)
        self.MLPy_Adapter = Adapter(adapter_dim, skip_connect=False)
        self.Img_Adapter = Adapter(adapter_dim)
        self.DSM_Adapter = Adapter(adapter_dim)
        self.scale = scale
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer)
        self.wx_Adapter = nn.Parameter(torch.full((1,), 0.5))
        self.wy_Adapter = nn.Parameter(torch.full((1,), 0.5))
        self.window_size = window_size
        self.num_heads = num_heads

    def forward_stacked(self, h: torch.Tensor) -> torch.Tensor:
        h = _AdapterAttnBranch.apply(h, self, _geom(h, self.num_heads, self.window_size),
                                     *_ad_params(self.Img_Adapter), *_ad_params(self.DSM_Adapter))
        return _AdapterMlpBranch.apply(h, self, *_ad_params(self.MLPx_Adapter), *_ad_params(self.MLPy_Adapter),
                                       self.wx_Adapter, self.wy_Adapter)

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        h = self.forward_stacked(torch.stack([x, y]).float().contiguous())
        return h[0], h[1]
