"""Thin tensor-level wrappers over the C ABI (pointer marshalling only, no arithmetic).

Every function enqueues on the current torch stream of the tensors' device and returns torch
tensors allocated by torch's caching allocator. CPU tensors are rejected loudly: the MFNet hot path
has no CPU / eager fallback (the CPU restatement lives in oracle/ and is test infrastructure).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import BF16, F32, MfnetError, call

_DT = {torch.float32: F32, torch.bfloat16: BF16}

# bench.py sets this to a list to time every GEMM launch with a pair of events recorded on the stream
# the kernel is launched on: entries are (start_event, end_event, algorithmic_flops).
GEMM_TRACE = None


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise MfnetError("MFNet HIP op called with a CPU tensor: the hot path runs only on a ROCm "
                             "device (no CPU fallback). Move the model/inputs to 'cuda'.")


def _rows2d(t: torch.Tensor):
    """(ptr-tensor, rows, cols, ld) for a tensor whose last dim is contiguous and whose leading
    dims collapse to a single stride."""
    assert t.stride(-1) == 1
    t2 = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t
    assert t2.dim() == 2
    return t2, t2.shape[0], t2.shape[1], t2.stride(0)


# ------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps, out_dtype=torch.bfloat16, want_stats=True):
    _chk(x, gamma, beta)
    x2, rows, C, ldx = _rows2d(x)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    call("mfnet_layernorm_fwd", x2.data_ptr(), _DT[x.dtype], ldx, gamma.data_ptr(), beta.data_ptr(),
         y.data_ptr(), _DT[out_dtype], C, _p(mean), _p(rstd), rows, C, float(eps), _stream(x))
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, want_f32=True, want_bf16=False, dgamma=None, dbeta=None):
    """dx = LN'(dy) [+ dres]; returns (dx_f32 | None, dx_bf16 | None). dgamma/dbeta are accumulated into."""
    _chk(dy, x, mean, rstd, gamma, dres)
    dy2, rows, C, lddy = _rows2d(dy)
    x2, _, _, ldx = _rows2d(x)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    dxb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    if dres is not None:
        assert dres.is_contiguous() and dres.dtype == torch.float32
    call("mfnet_layernorm_bwd", dy2.data_ptr(), _DT[dy.dtype], lddy, x2.data_ptr(), _DT[x.dtype], ldx,
         mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), _p(dres), _p(dx), _p(dxb), C,
         _p(dgamma), _p(dbeta), rows, C, _stream(x))
    return dx, dxb


# ------------------------------------------------------------------------------------------ GEMM
def gemm(a, w, *, bias=None, U=None, V=None, rows_per_group=0, n_groups=1, lr_scale=1.0, act=0, aux=None,
         want_pre=False, residual=None, res_row_mod=0, out=None, out_dtype=torch.bfloat16):
    """out = a @ w^T (+bias +U V_g^T) -> act -> (+residual). a [M,K] bf16, w [N,K] bf16.
    V: [G,N,R] fp32. act: 0 none, 1 gelu (want_pre also returns the pre-activation), 2 multiply by
    gelu'(aux). Returns out or (out, pre)."""
    _chk(a, w, bias, U, V, aux, residual, out)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    a2, M, K, lda = _rows2d(a)
    N, K2 = w.shape
    assert K2 == K and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if want_pre else None
    R = 0
    vgs = 0
    if U is not None:
        assert U.dtype == torch.float32 and V.dtype == torch.float32 and U.is_contiguous() and V.is_contiguous()
        R = U.shape[-1]
        assert V.shape[-1] == R and V.shape[-2] == N
        vgs = V.stride(0) if V.dim() == 3 else 0
    if GEMM_TRACE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(a.device))
    call("mfnet_gemm_bf16", a2.data_ptr(), lda, w.data_ptr(), w.stride(0), M, N, K, _p(bias),
         _p(U), R if U is not None else 0, _p(V), vgs, rows_per_group, n_groups, R, float(lr_scale), act,
         _p(aux), aux.stride(-2) if aux is not None else 0, _p(pre), N if pre is not None else 0,
         _p(residual), residual.stride(-2) if residual is not None else 0, res_row_mod,
         out.data_ptr(), _DT[out.dtype], out.stride(-2), _stream(a))
    if GEMM_TRACE is not None:
        ev1.record(torch.cuda.current_stream(a.device))
        GEMM_TRACE.append((ev0, ev1, 2.0 * M * N * K))
    return (out, pre) if want_pre else out


def skinny_nt(x, wt, scale=1.0, out=None):
    """out[m, j] = scale * sum_k x[m,k] wt[j,k]; x bf16 [M,K], wt fp32 [R,K] -> fp32 [M,R]."""
    _chk(x, wt, out)
    x2, M, K, ldx = _rows2d(x)
    R = wt.shape[0]
    assert wt.dtype == torch.float32 and wt.stride(1) == 1 and wt.shape[1] == K
    if out is None:
        out = torch.empty((M, R), dtype=torch.float32, device=x.device)
    call("mfnet_skinny_nt", x2.data_ptr(), ldx, wt.data_ptr(), wt.stride(0), out.data_ptr(), out.stride(0),
         M, K, R, float(scale), _stream(x))
    return out


def skinny_tn(y, u, transposed: bool, scale=1.0, out=None):
    """out[c, j] (or [j, c] if transposed) += scale * sum_m y[m,c] u[m,j]; y bf16 [M,C], u fp32 [M,R]."""
    _chk(y, u, out)
    y2, M, C, ldy = _rows2d(y)
    R = u.shape[1]
    assert u.dtype == torch.float32 and u.stride(1) == 1 and u.shape[0] == M
    if out is None:
        out = torch.zeros((R, C) if transposed else (C, R), dtype=torch.float32, device=y.device)
    assert out.is_contiguous()
    call("mfnet_skinny_tn", y2.data_ptr(), ldy, u.data_ptr(), u.stride(0), out.data_ptr(), M, C, R,
         1 if transposed else 0, float(scale), _stream(y))
    return out


def cast(x, dtype):
    _chk(x)
    assert x.is_contiguous() and x.numel() % 8 == 0
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("mfnet_cast", x.data_ptr(), _DT[x.dtype], y.data_ptr(), _DT[dtype], x.numel(), _stream(x))
    return y


def im2col_patch(rgb, dsm, patch=16):
    """rgb [B,3,H,W] fp32, dsm [B,H,W] fp32 -> [2*B*(H/P)*(W/P), 3*P*P] bf16 (rgb rows first)."""
    _chk(rgb, dsm)
    assert rgb.is_contiguous() and dsm.is_contiguous() and rgb.dtype == torch.float32 and dsm.dtype == torch.float32
    B, _, H, W = rgb.shape
    cols = torch.empty((2 * B * (H // patch) * (W // patch), 3 * patch * patch), dtype=torch.bfloat16, device=rgb.device)
    call("mfnet_im2col_patch", rgb.data_ptr(), dsm.data_ptr(), cols.data_ptr(), B, H, W, patch, _stream(rgb))
    return cols


def im2col3x3(x, G, H, W):
    _chk(x)
    C = x.shape[-1]
    assert x.is_contiguous() and x.dtype == torch.bfloat16
    cols = torch.empty((G * H * W, 9 * C), dtype=torch.bfloat16, device=x.device)
    call("mfnet_im2col3x3", x.data_ptr(), cols.data_ptr(), G, H, W, C, _stream(x))
    return cols


def col2im3x3(dcols, G, H, W, C):
    _chk(dcols)
    assert dcols.is_contiguous() and dcols.dtype == torch.bfloat16
    dx = torch.empty((G * H * W, C), dtype=torch.float32, device=dcols.device)
    call("mfnet_col2im3x3", dcols.data_ptr(), dx.data_ptr(), G, H, W, C, _stream(dcols))
    return dx


def lora_mix_fwd(a, wx, wy):
    _chk(a, wx, wy)
    M, r = a.shape
    U = torch.empty((M, 2 * r), dtype=torch.float32, device=a.device)
    call("mfnet_lora_mix_fwd", a.data_ptr(), wx.data_ptr(), wy.data_ptr(), U.data_ptr(), M, r, _stream(a))
    return U


def lora_mix_bwd(P, a, wx, wy):
    _chk(P, a, wx, wy)
    M, r = a.shape
    da = torch.empty_like(a)
    dw = torch.zeros(2, dtype=torch.float32, device=a.device)
    call("mfnet_lora_mix_bwd", P.data_ptr(), a.data_ptr(), wx.data_ptr(), wy.data_ptr(), da.data_ptr(),
         dw.data_ptr(), M, r, _stream(a))
    return da, dw


# ------------------------------------------------------------------------------------------ attention
def attn_fwd(qkv, qkv_bias, tbl, G, Sh, Sw, heads, ws, Lh, Lw, scale):
    """qkv [G*Sh*Sw, 3D] bf16 -> (out [G*Sh*Sw, D] bf16, lse [G, heads, Sh*Sw] fp32)."""
    _chk(qkv, qkv_bias, tbl)
    D = qkv.shape[1] // 3
    hd = D // heads
    assert qkv.is_contiguous() and tbl.is_contiguous() and tbl.dtype == torch.bfloat16
    out = torch.empty((qkv.shape[0], D), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((G, heads, Sh * Sw), dtype=torch.float32, device=qkv.device)
    call("mfnet_attn_fwd", qkv.data_ptr(), qkv_bias.data_ptr(), tbl.data_ptr(), out.data_ptr(), lse.data_ptr(),
         G, Sh, Sw, heads, hd, ws, Lh, Lw, tbl.shape[0], float(scale), _stream(qkv))
    return out, lse


def attn_bwd(qkv, qkv_bias, tbl, tblT, out, dout, lse, G, Sh, Sw, heads, ws, Lh, Lw, scale):
    _chk(qkv, qkv_bias, tbl, tblT, out, dout, lse)
    D = qkv.shape[1] // 3
    hd = D // heads
    assert dout.is_contiguous() and dout.dtype == torch.bfloat16
    dqkv = torch.empty_like(qkv)
    call("mfnet_attn_bwd", qkv.data_ptr(), qkv_bias.data_ptr(), tbl.data_ptr(), tblT.data_ptr(), out.data_ptr(),
         dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), G, Sh, Sw, heads, hd, ws, Lh, Lw, tbl.shape[0],
         float(scale), _stream(qkv))
    return dqkv


# ------------------------------------------------------------------------------------------ train / infer glue
def ce_loss_fwd(logits, target, weight=None, ignore_index=-100, want_pred=False):
    _chk(logits, target, weight)
    assert logits.is_contiguous() and logits.dtype == torch.float32 and target.dtype == torch.int64 and target.is_contiguous()
    B, C = logits.shape[:2]
    HW = logits.numel() // (B * C)
    acc = torch.zeros(4, dtype=torch.float32, device=logits.device)   # sum w*nll, sum w, #correct, #valid
    pred = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.uint8, device=logits.device) if want_pred else None
    call("mfnet_ce_loss_fwd", logits.data_ptr(), target.data_ptr(), _p(weight), acc.data_ptr(), _p(pred),
         B, C, HW, ignore_index, _stream(logits))
    return acc, pred


def ce_loss_bwd(logits, target, acc, weight=None, ignore_index=-100, gscale=1.0):
    B, C = logits.shape[:2]
    HW = logits.numel() // (B * C)
    d = torch.empty_like(logits)
    call("mfnet_ce_loss_bwd", logits.data_ptr(), target.data_ptr(), _p(weight), acc.data_ptr(), d.data_ptr(),
         float(gscale), B, C, HW, ignore_index, _stream(logits))
    return d


def sgd_flat(p, g, buf, lr, momentum, wd, gscale=1.0):
    _chk(p, g, buf)
    assert p.is_contiguous() and g.is_contiguous() and p.numel() == g.numel()
    call("mfnet_sgd_flat", p.data_ptr(), g.data_ptr(), _p(buf), p.numel(), float(lr), float(momentum), float(wd),
         float(gscale), _stream(p))


def tile_accumulate(pred, logits, coords, atomic=True):
    """pred [C,H,W] fp32 += logits [n,C,h,w] placed at coords [n,2] int32 (row, col)."""
    _chk(pred, logits, coords)
    assert pred.is_contiguous() and logits.is_contiguous() and coords.dtype == torch.int32 and coords.is_contiguous()
    C, H, W = pred.shape
    n, C2, h, w = logits.shape
    assert C2 == C
    call("mfnet_tile_accumulate", pred.data_ptr(), logits.data_ptr(), coords.data_ptr(), n, C, H, W, h, w,
         1 if atomic else 0, _stream(pred))


def argmax_chw(pred):
    _chk(pred)
    C = pred.shape[0]
    out = torch.empty(pred.shape[1:], dtype=torch.uint8, device=pred.device)
    call("mfnet_argmax_chw", pred.data_ptr(), out.data_ptr(), C, out.numel(), _stream(pred))
    return out


def minmax(x):
    _chk(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    mm = torch.empty(2, dtype=torch.float32, device=x.device)
    call("mfnet_minmax", x.data_ptr(), mm.data_ptr(), x.numel(), _stream(x))
    return mm


def scale_shift(x, mm, mode=0):
    y = torch.empty_like(x)
    call("mfnet_scale_shift", x.data_ptr(), y.data_ptr(), mm.data_ptr(), 0.0, x.numel(), mode, _stream(x))
    return y


def adapter_mix(h, ada, wx, wy, out=None):
    _chk(h, ada, wx, wy)
    M, D = h.shape
    if out is None:
        out = torch.empty_like(h)
    call("mfnet_adapter_mix", h.data_ptr(), ada.data_ptr(), wx.data_ptr(), wy.data_ptr(), out.data_ptr(), None, M, D, _stream(h))
    return out


def adapter_mix_bwd(dh, ada, wx, wy):
    M, D = dh.shape
    dada = torch.empty((M, D), dtype=torch.bfloat16, device=dh.device)
    dw = torch.zeros(2, dtype=torch.float32, device=dh.device)
    call("mfnet_adapter_mix_bwd", dh.data_ptr(), ada.data_ptr(), wx.data_ptr(), wy.data_ptr(), dada.data_ptr(),
         dw.data_ptr(), None, M, D, _stream(dh))
    return dada, dw


def bias_grad(dy, out=None):
    _chk(dy)
    dy2, M, N, ld = _rows2d(dy)
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=dy.device)
    call("mfnet_bias_grad", dy2.data_ptr(), ld, out.data_ptr(), M, N, _stream(dy))
    return out


def add_bf16(a, b, out):
    _chk(a, b, out)
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
    call("mfnet_add_bf16", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream(a))
    return out


def transpose_pad(x, mult=64):
    """x [M,N] bf16 -> [N, Mp] bf16, Mp = M rounded up to `mult`, zero padded."""
    _chk(x)
    x2, M, N, ld = _rows2d(x)
    Mp = (M + mult - 1) // mult * mult
    y = torch.empty((N, Mp), dtype=torch.bfloat16, device=x.device)
    call("mfnet_transpose_bf16", x2.data_ptr(), ld, y.data_ptr(), M, N, Mp, _stream(x))
    return y


def gemm_tn(dy, x):
    """dW[n,k] = sum_m dy[m,n] x[m,k]  (fp32 [N,K]); both bf16 row-major with M leading."""
    return gemm(transpose_pad(dy), transpose_pad(x), out_dtype=torch.float32)
