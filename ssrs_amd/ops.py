"""Thin tensor-level wrappers over the C ABI (pointer marshalling only, no arithmetic).

Every function enqueues on the current torch stream of the tensors' device and returns torch
tensors allocated by torch's caching allocator. CPU tensors are rejected loudly: the MFNet hot path
has no CPU / eager fallback (the CPU restatement lives in oracle/ and is test infrastructure).
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import _lib
from ._lib import BF16, F32, MfnetError, call

_DT = {torch.float32: F32, torch.bfloat16: BF16}

# bench.py sets this to a list to time every GEMM launch with a pair of events recorded on the stream
# the kernel is launched on: entries are (start_event, end_event, algorithmic_flops).
GEMM_TRACE = None
# likewise for the dominant HBM-bound kernel (ln_bwd_kernel): entries are (start_event, end_event, algorithmic_bytes)
LN_BWD_TRACE = None
# KERNEL_TRACE: when a dict, every entry point launched through tcall() is bracketed by an event pair on its stream and
# appends (ev0, ev1, algorithmic bytes, algorithmic flops) under its name - bench.py's roofline tables (the HBM-bound kernels
# against 8 TB/s, attention against the MFMA peak). None (the default) costs one comparison per call.
KERNEL_TRACE = None


# fp32 training (the reference trains in fp32: trainer.py:146-159). While FP32 is set, every request for a bf16 activation,
# gradient or weight image is answered in fp32 (_odt) and the wrappers below route fp32 operands of the bf16-only MFMA kernels to
# mfnet_gemm_f32 and the fp32 twins (attention, GLA attention, up-sampling, adapter mix). The module backward passes (vit.py,
# vit_adapter.py, decoder_ops.py) are the same code in both modes; autograd Functions remember the mode of their forward
# (`keeps_precision`), so a backward pass runs in the mode its forward ran in.
FP32 = False


def _odt(dt):
    return torch.float32 if (FP32 and dt == torch.bfloat16) else dt


class precision:
    """`with ops.precision("fp32"):` - the calls inside compute in fp32 (see FP32)."""

    def __init__(self, name):
        if name not in ("bf16", "fp32"):
            raise ValueError(f"precision {name!r}: 'bf16' (default) or 'fp32'")
        self.want = name == "fp32"

    def __enter__(self):
        global FP32
        self.prev, FP32 = FP32, self.want
        return self

    def __exit__(self, *exc):
        global FP32
        FP32 = self.prev
        return False


def keeps_precision(cls):
    """class decorator for torch.autograd.Function: the backward runs in the precision mode its forward ran in"""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._mfnet_fp32 = FP32
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        global FP32
        prev, FP32 = FP32, ctx._mfnet_fp32
        try:
            return bwd(ctx, *grads)
        finally:
            FP32 = prev
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def tcall(name, nbytes, flops, *args):
    """call(name, *args) with its algorithmic bytes / flops noted for KERNEL_TRACE (each operand counted once)."""
    if KERNEL_TRACE is None:
        return call(name, *args)
    st = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(st)
    r = call(name, *args)
    ev1.record(st)
    KERNEL_TRACE.setdefault(name, []).append((ev0, ev1, float(nbytes), float(flops)))
    return r


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


def gemm_f32(a, w, bias=None, act=0, residual=None, res_row_mod=0, alpha=1.0, out=None, accumulate=False):
    """out = act(alpha * a @ w^T + bias) (+ residual) (+ out if accumulate). a [M,K], w [N,K] fp32 (unit inner stride, row
    strides multiples of 4); out [M,N] fp32 (any row stride). act: 0 none, 1 GELU(erf), 3 ReLU6, 6 ReLU."""
    _chk(a, w, bias, residual, out)
    assert a.dtype == torch.float32 and w.dtype == torch.float32 and a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        assert not accumulate
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(1) == 1 and residual.shape[1] == N
    call("mfnet_gemm_f32", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K, _p(bias), float(alpha), int(act),
         _p(residual), residual.stride(0) if residual is not None else 0, int(res_row_mod), out.data_ptr(), out.stride(0),
         1 if accumulate else 0, _stream(a))
    return out


def _t32(x):
    """contiguous transpose of a 2-D fp32 view (a copy: the fp32 GEMM reads both operands k-major)"""
    return x.t().contiguous()


def _gemm_route32(a, w, bias, U, V, rows_per_group, n_groups, lr_scale, act, aux, want_pre, residual, res_row_mod, out, pre_out=None):
    """ops.gemm's contract on fp32 operands: a @ w^T (+ bias + lr_scale * U V_g^T) -> act -> (+ residual), as mfnet_gemm_f32 launches
    plus the elementwise kernels for the epilogues the fp32 GEMM does not fuse. act: 0, 1 GELU, 2 x gelu'(aux), 3 ReLU6,
    4 x relu6'(aux), 5 ReLU6 after the residual."""
    a2, M, K, lda = _rows2d(a)
    assert w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] == K and w.stride(1) == 1
    N = w.shape[0]
    if out is not None:
        assert out.dtype == torch.float32 and out.dim() == 2
    fused_res = residual is not None and act in (0, 3) and not want_pre and U is None
    direct = U is None and not want_pre and act in (0, 1, 3) and (residual is None or fused_res)
    if direct:                              # the fp32 GEMM's own epilogue: act, then the residual
        return gemm_f32(a2, w, bias=bias, act=act, residual=residual, res_row_mod=res_row_mod, out=out)
    lin_res = residual if (act in (0, 5) and not want_pre) else None           # the residual joins before anything non-linear happens
    plain = act == 0 and not want_pre
    pre = gemm_f32(a2, w, bias=bias, residual=lin_res, res_row_mod=res_row_mod if lin_res is not None else 0,
                   out=out if plain else pre_out)
    if U is not None:
        G = n_groups if V.dim() == 3 else 1
        for g in range(G):
            rows = slice(g * rows_per_group, (g + 1) * rows_per_group) if G > 1 else slice(0, M)
            gemm_f32(U[rows], V[g] if V.dim() == 3 else V, alpha=lr_scale, out=pre[rows], accumulate=True)
    if plain:
        return pre
    if act == 0:                            # want_pre without an activation: pre, and pre + residual
        res = add3(pre, residual, out=out) if residual is not None else pre
        return res, pre
    if act == 1:
        res = gelu_fwd(pre)
    elif act == 2:
        res = gelu_bwd(aux.contiguous(), pre)
    elif act in (3, 5):
        res = _relu6(pre)
    elif act == 4:
        res = act_bwd(pre, aux.contiguous(), ACT_RELU6, out_dtype=torch.float32)
    else:
        raise MfnetError(f"fp32 GEMM route: activation code {act}")
    if residual is not None and act != 5:
        res = add3(res, residual)
    if out is not None:
        out.copy_(res)
        res = out
    return (res, pre) if want_pre else res


# ------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps, out_dtype=torch.bfloat16, want_stats=True):
    _chk(x, gamma, beta)
    out_dtype = _odt(out_dtype)
    x2, rows, C, ldx = _rows2d(x)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    tcall("mfnet_layernorm_fwd", rows * C * (x.element_size() + y.element_size()) + rows * 8, 0,
          x2.data_ptr(), _DT[x.dtype], ldx, gamma.data_ptr(), beta.data_ptr(),
         y.data_ptr(), _DT[out_dtype], C, _p(mean), _p(rstd), rows, C, float(eps), _stream(x))
    return y, mean, rstd


def layernorm_fwd_lora(x, gamma, beta, eps, A, wx=None, wy=None):
    """LayerNorm -> bf16 plus the LoRA down-projection a = y @ A_g^T in the same launch. A fp32 [G, R, C] (G = 2: the first half
    of the rows is modality x). With wx / wy also the mixed factor U [rows, 2R] of lora.Linear. -> (y, mean, rstd, a, U | None)."""
    _chk(x, gamma, beta, A, wx, wy)
    x2, rows, C, ldx = _rows2d(x)
    assert A.dim() == 3 and A.is_contiguous() and A.dtype == torch.float32 and A.shape[2] == C
    G, R = A.shape[0], A.shape[1]
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    a = torch.empty((rows, R), dtype=torch.float32, device=x.device)
    U = torch.empty((rows, 2 * R), dtype=torch.float32, device=x.device) if wx is not None else None
    tcall("mfnet_layernorm_fwd_lora", rows * C * (x.element_size() + 2) + rows * (8 + 4 * R + (8 * R if U is not None else 0)), 0, x2.data_ptr(), _DT[x.dtype], ldx, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), C, mean.data_ptr(),
         rstd.data_ptr(), rows, C, float(eps), A.data_ptr(), R * C, G, R, a.data_ptr(), _p(wx), _p(wy), _p(U), _stream(x))
    return y, mean, rstd, a, U


def lora_down_mix(x, A, wx=None, wy=None):
    """a = x @ A_g^T for a bf16 token matrix x [rows, C] (A fp32 [G, R, C]; G = 2: first half of the rows is modality x) and, with
    wx / wy, the mixed factor U [rows, 2R] of lora.Linear, in one launch. -> (a, U | None)."""
    _chk(x, A, wx, wy)
    x2, rows, C, ldx = _rows2d(x)
    assert x.dtype == torch.bfloat16 and A.dim() == 3 and A.is_contiguous() and A.dtype == torch.float32 and A.shape[2] == C
    G, R = A.shape[0], A.shape[1]
    a = torch.empty((rows, R), dtype=torch.float32, device=x.device)
    U = torch.empty((rows, 2 * R), dtype=torch.float32, device=x.device) if wx is not None else None
    tcall("mfnet_lora_down_mix", rows * C * 2 + rows * (4 * R + (8 * R if U is not None else 0)), 0, x2.data_ptr(), ldx, rows, C, A.data_ptr(), R * C, G, R, a.data_ptr(), _p(wx), _p(wy), _p(U), _stream(x))
    return a, U


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, want_f32=True, want_bf16=False, dgamma=None, dbeta=None):
    """dx = LN'(dy) [+ dres]; returns (dx_f32 | None, dx_bf16 | None). dgamma/dbeta are accumulated into."""
    _chk(dy, x, mean, rstd, gamma, dres)
    if FP32 and want_bf16:                  # fp32 training: the "bf16 copy" of the gradient is the fp32 gradient itself
        dx, _ = layernorm_bwd(dy, x, mean, rstd, gamma, dres, True, False, dgamma, dbeta)
        return (dx if want_f32 else None), dx
    dy2, rows, C, lddy = _rows2d(dy)
    x2, _, _, ldx = _rows2d(x)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    dxb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    if dres is not None:
        assert dres.is_contiguous() and dres.dtype == torch.float32
    # weight gradients: one deterministic column reduction when the rows are dense (the dx kernel's own path adds
    # them with float atomics from every wave)
    split_w = (dgamma is not None and lddy == C and ldx == C and C % 4 == 0 and C <= 1024
               and dy2.is_contiguous() and x2.is_contiguous())
    if LN_BWD_TRACE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(x.device))
    per_el = dy.element_size() + x.element_size() + (4 if dres is not None else 0) + (4 if want_f32 else 0) + (2 if want_bf16 else 0)
    tcall("mfnet_layernorm_bwd", rows * C * per_el + rows * 8, 0, dy2.data_ptr(), _DT[dy.dtype], lddy, x2.data_ptr(), _DT[x.dtype], ldx,
         mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), _p(dres), _p(dx), _p(dxb), C,
         None if split_w else _p(dgamma), None if split_w else _p(dbeta), rows, C, _stream(x))
    if LN_BWD_TRACE is not None:
        ev1.record(torch.cuda.current_stream(x.device))
        # every operand is streamed once: dy, x (+ the residual gradient) in, dx (fp32 and / or bf16) out, the row statistics
        LN_BWD_TRACE.append((ev0, ev1, rows * C * per_el + rows * 8, (rows, C)))
    if split_w:
        ws, tk = _colreduce_ws(1, rows, C, x.device)
        call("mfnet_layernorm_bwd_weights", dy2.data_ptr(), _DT[dy.dtype], x2.data_ptr(), _DT[x.dtype], mean.data_ptr(), rstd.data_ptr(),
             dgamma.data_ptr(), dbeta.data_ptr(), rows, C, ws.data_ptr(), tk.data_ptr(), _stream(x))
    return dx, dxb


# ------------------------------------------------------------------------------------------ GEMM
class ZeroArena:
    """Zero-initialised fp32 scratch outputs of one train step, carved out of ONE buffer that is cleared by a
    single fill at the start of the step (the step has ~250 small "adds into its output" results; a fill
    launch each costs ~1 ms per step). The first step only records the demand; later steps bump-allocate.
    Tensors taken from the arena are valid until the next step begins."""
    current = None

    def __init__(self):
        self.buf, self.off, self.need = None, 0, 0

    def __enter__(self):
        ZeroArena.current = self
        self.off = 0
        if self.buf is not None:
            self.buf.zero_()
        return self

    def __exit__(self, *exc):
        ZeroArena.current = None
        if self.buf is None and self.need:
            self.buf = torch.empty(int(self.need * 1.25) + 4096, dtype=torch.float32, device=self.dev)
        return False

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        span = (n + 63) // 64 * 64
        if self.buf is None or self.off + span > self.buf.numel() or self.buf.device != device:
            self.need += span
            self.dev = device
            return torch.zeros(shape, dtype=torch.float32, device=device)
        t = self.buf[self.off:self.off + n].view(shape)
        self.off += span
        return t


def zeros(shape, device):
    """fp32 zeros: from the active step arena if there is one, else a fresh tensor."""
    if isinstance(shape, int):
        shape = (shape,)
    a = ZeroArena.current
    return a.take(tuple(shape), device) if a is not None else torch.zeros(shape, dtype=torch.float32, device=device)


def gemm(a, w, *, bias=None, U=None, V=None, rows_per_group=0, n_groups=1, lr_scale=1.0, act=0, aux=None,
         want_pre=False, residual=None, res_row_mod=0, out=None, out_dtype=torch.bfloat16, k_splits=1, tile=0):
    """out = a @ w^T (+bias +U V_g^T) -> act -> (+residual). a [M,K] bf16, w [N,K] bf16.
    V: [G,N,R] fp32. act: 0 none, 1 gelu (want_pre also returns the pre-activation), 2 multiply by
    gelu'(aux). Returns out or (out, pre)."""
    _chk(a, w, bias, U, V, aux, residual, out)
    if a.dtype == torch.float32:
        assert k_splits == 1 and not tile
        if U is not None and U.shape[-1] % 4:
            padc = 4 - U.shape[-1] % 4
            U, V = torch.nn.functional.pad(U, (0, padc)), torch.nn.functional.pad(V, (0, padc))
        return _gemm_route32(a, w, bias, U, V, rows_per_group, n_groups, lr_scale, act, aux, want_pre, residual, res_row_mod, out)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    a2, M, K, lda = _rows2d(a)
    N, K2 = w.shape
    assert K2 == K and w.stride(1) == 1
    if k_splits > 1:
        assert out is None and out_dtype == torch.float32
        out = torch.empty((k_splits, M, N), dtype=torch.float32, device=a.device)      # per-split partials
    elif out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if want_pre else None
    R = 0
    vgs = 0
    if U is not None:
        if U.shape[-1] % 4:          # odd LoRA ranks / rank 2: zero columns up to a multiple of 4 (small copies of the thin factors)
            padc = 4 - U.shape[-1] % 4
            U, V = torch.nn.functional.pad(U, (0, padc)), torch.nn.functional.pad(V, (0, padc))
        assert U.dtype == torch.float32 and V.dtype == torch.float32 and U.is_contiguous() and V.is_contiguous()
        R = U.shape[-1]
        assert V.shape[-1] == R and V.shape[-2] == N
        vgs = V.stride(0) if V.dim() == 3 else 0
    if GEMM_TRACE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(a.device))
    args = (a2.data_ptr(), lda, w.data_ptr(), w.stride(0), M, N, K, _p(bias),
            _p(U), R if U is not None else 0, _p(V), vgs, rows_per_group, n_groups, R, float(lr_scale), act,
            _p(aux), aux.stride(-2) if aux is not None else 0, _p(pre), N if pre is not None else 0,
            _p(residual), residual.stride(-2) if residual is not None else 0, res_row_mod,
            out.data_ptr(), _DT[out.dtype], out.stride(-2), k_splits)
    if tile:         # the caller names the tile shape (tests / benchmarks): -1 = 128x128 kernel, 3/4/6/8/9 = 256 x 32*tile
        call("mfnet_gemm_bf16_tiled", *args, int(tile), _stream(a))
    else:
        call("mfnet_gemm_bf16", *args, _stream(a))
    if GEMM_TRACE is not None:
        ev1.record(torch.cuda.current_stream(a.device))
        GEMM_TRACE.append((ev0, ev1, 2.0 * M * N * K, (M, N, K)))
    return (out, pre) if want_pre else out


def gemm_grouped(a, w, *, bias=None, rows_per_group, act=0, aux=None, want_pre=False, residual=None, out=None,
                 out_dtype=torch.bfloat16):
    """Per-group weights: a [M,K] bf16, w [G,N,K] bf16, bias [G,N] fp32 | None; rows [g*rpg, (g+1)*rpg) use w[g], bias[g]
    (the per-modality adapters in one launch). Otherwise as gemm (no low-rank term)."""
    _chk(a, w, bias, aux, residual, out)
    if a.dtype == torch.float32:
        a2, M, K, lda = _rows2d(a)
        G, N = w.shape[0], w.shape[1]
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        pre = torch.empty((M, N), dtype=torch.float32, device=a.device) if want_pre else None
        for g in range(G):
            rows = slice(g * rows_per_group, (g + 1) * rows_per_group)
            _gemm_route32(a2[rows], w[g], bias[g] if bias is not None else None, None, None, 0, 1, 1.0, act,
                          aux[rows] if aux is not None else None, want_pre, residual[rows] if residual is not None else None, 0,
                          out[rows], pre[rows] if want_pre else None)
        return (out, pre) if want_pre else out
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.dim() == 3 and w.is_contiguous()
    a2, M, K, lda = _rows2d(a)
    G, N, K2 = w.shape
    assert K2 == K and (bias is None or (bias.is_contiguous() and tuple(bias.shape) == (G, N)))
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if want_pre else None
    if GEMM_TRACE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(a.device))
    call("mfnet_gemm_bf16_grouped", a2.data_ptr(), lda, w.data_ptr(), K, N * K, M, N, K, _p(bias), N if bias is not None else 0,
         rows_per_group, G, act, _p(aux), aux.stride(-2) if aux is not None else 0, _p(pre), N if pre is not None else 0,
         _p(residual), residual.stride(-2) if residual is not None else 0, out.data_ptr(), _DT[out.dtype], out.stride(-2), _stream(a))
    if GEMM_TRACE is not None:
        ev1.record(torch.cuda.current_stream(a.device))
        GEMM_TRACE.append((ev0, ev1, 2.0 * M * N * K, (M, N, K)))
    return (out, pre) if want_pre else out


def skinny_nt(x, wt, scale=1.0, out=None, rows_per_group=0):
    """out[m, j] = scale * sum_k x[m,k] wt[g(m)][j,k]; x bf16 [M,K]; wt fp32 [R,K] or [G,R,K] with
    g(m) = m // rows_per_group (per-modality matrices) -> fp32 [M, R] (row stride a multiple of 4). The kernel holds up to
    16 output columns per row tile; wider factors (LoRA ranks 12 / 16: stacked factors of 24 / 32 rows) go in column chunks."""
    _chk(x, wt, out)
    x2, M, K, ldx = _rows2d(x)
    G = wt.shape[0] if wt.dim() == 3 else 1
    R = wt.shape[-2]
    assert wt.dtype == torch.float32 and wt.stride(-1) == 1 and wt.shape[-1] == K
    if out is None:
        Rp = (R + 3) // 4 * 4
        out = torch.empty((M, Rp), dtype=torch.float32, device=x.device)[:, :R]
    if x.dtype == torch.float32:
        for g in range(G):
            rows = slice(g * rows_per_group, (g + 1) * rows_per_group) if G > 1 else slice(0, M)
            gemm_f32(x2[rows], wt[g] if G > 1 else wt.reshape(R, K), alpha=scale, out=out[rows])
        return out
    for c0 in range(0, R, 16):
        c1 = min(R, c0 + 16)
        wc, oc = wt[..., c0:c1, :], out[:, c0:c1]
        call("mfnet_skinny_nt", x2.data_ptr(), ldx, wc.data_ptr(), wc.stride(-2), wc.stride(0) if G > 1 else 0, rows_per_group if G > 1 else 0, G,
             oc.data_ptr(), oc.stride(0), M, K, c1 - c0, float(scale), _stream(x))
    return out


class SplitSum:
    """scale * sum over dim 0 of `part`, not taken yet: `bank.deliver` folds the sum into its gather launch (the
    partial tiles of mfnet_skinny_tn / mfnet_gemm_tn_bf16), so weight gradients that go straight into the flat
    gradient buffer need no reduce kernel of their own. Supports the view operations the backward passes apply to a
    weight gradient (indexing, view, permute, t); `materialize()` gives the ordinary tensor."""

    def __init__(self, part, scale=1.0):
        self.part, self.scale = part, float(scale)

    shape = property(lambda self: self.part.shape[1:])
    device = property(lambda self: self.part.device)
    dtype = property(lambda self: self.part.dtype)

    def dim(self):
        return self.part.dim() - 1

    def __getitem__(self, idx):
        idx = idx if isinstance(idx, tuple) else (idx,)
        return SplitSum(self.part[(slice(None),) + idx], self.scale)

    def view(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        return SplitSum(self.part.view(self.part.shape[0], *shape), self.scale)

    def permute(self, *dims):
        dims = dims[0] if len(dims) == 1 and isinstance(dims[0], (tuple, list)) else dims
        return SplitSum(self.part.permute(0, *[d + 1 for d in dims]), self.scale)

    def t(self):
        assert self.dim() == 2
        return SplitSum(self.part.transpose(1, 2), self.scale)

    def materialize(self):
        if TN_QUEUE:
            tn_flush()              # the partial tiles may belong to a product that is still collected
        out = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.part.device)
        gather([(out, self)])
        return out


def _tn_chunks(R):
    """column chunks of the widths mfnet_skinny_tn is built for (16 / 8 / 4) covering R (a multiple of 4)"""
    out, c0 = [], 0
    while c0 < R:
        w = 16 if R - c0 >= 16 else (8 if R - c0 >= 8 else 4)
        out.append((c0, c0 + w))
        c0 += w
    return out


def skinny_tn(y, u, transposed: bool, scale=1.0, out=None, n_groups=1, lazy=False):
    """per group z (rows [z*M/G, (z+1)*M/G)): out[z][c, j] (or [j, c] if transposed) += scale * sum_m y[m,c] u[m,j];
    y bf16 [M,C], u fp32 [M,R]. Returns [G, C, R] / [G, R, C] (the leading dim is dropped for n_groups == 1).
    lazy: returns a SplitSum over the partial tiles instead (no reduce launch; for gradients handed to deliver).
    R other than 4 / 8 / 16 (LoRA ranks 12 / 16: factors of 12, 24 or 32 columns) runs in column chunks and always returns
    the dense tensor."""
    _chk(y, u, out)
    y2, M, C, ldy = _rows2d(y)
    R = u.shape[1]
    assert u.dtype == torch.float32 and u.stride(1) == 1 and u.shape[0] == M and M % n_groups == 0
    if y.dtype == torch.float32:
        Mg = M // n_groups
        res = out if out is not None else torch.empty((n_groups, R, C) if transposed else (n_groups, C, R), dtype=torch.float32, device=y.device)
        r3 = res if res.dim() == 3 else res[None]
        for g in range(n_groups):
            yt, ut = _t32(y2[g * Mg:(g + 1) * Mg]), _t32(u[g * Mg:(g + 1) * Mg])
            gemm_f32(ut, yt, alpha=scale, out=r3[g], accumulate=out is not None) if transposed else \
                gemm_f32(yt, ut, alpha=scale, out=r3[g], accumulate=out is not None)
        return res[0] if (n_groups == 1 and res.dim() == 3) else res
    if R % 4:                # LoRA rank 2: zero columns up to the next width the kernel is built for (a [M, 4] copy)
        assert out is None
        res = skinny_tn(y, torch.nn.functional.pad(u, (0, 4 - R % 4)), transposed, scale, None, n_groups, False)
        return res[..., :R, :] if transposed else res[..., :R]
    if R not in (4, 8, 16):
        assert out is None
        parts = [skinny_tn(y, u[:, c0:c1], transposed, scale, None, n_groups, False) for c0, c1 in _tn_chunks(R)]
        return torch.cat(parts, -2 if transposed else -1)
    Mg = M // n_groups
    nws = _lib.lib().mfnet_skinny_tn_workspace_floats(Mg, C, R, n_groups)
    ws = torch.empty(nws, dtype=torch.float32, device=y.device)
    if lazy:
        assert out is None
        call("mfnet_skinny_tn", y2.data_ptr(), ldy, u.data_ptr(), u.stride(0), None, Mg, C, R,
             1 if transposed else 0, float(scale), n_groups, C * R, ws.data_ptr(), nws, _stream(y))
        part = ws.view(n_groups, nws // (n_groups * C * R), C, R).transpose(0, 1)          # [splits, G, C, R]
        if transposed:
            part = part.transpose(2, 3)
        res = SplitSum(part, scale)
        return res[0] if n_groups == 1 else res
    if out is None:
        out = zeros((n_groups, R, C) if transposed else (n_groups, C, R), y.device)
    assert out.is_contiguous()
    call("mfnet_skinny_tn", y2.data_ptr(), ldy, u.data_ptr(), u.stride(0), out.data_ptr(), Mg, C, R,
         1 if transposed else 0, float(scale), n_groups, C * R, ws.data_ptr(), nws, _stream(y))
    return out[0] if (n_groups == 1 and out.dim() == 3) else out


def cast(x, dtype):
    _chk(x)
    dtype = _odt(dtype)
    if x.dtype == dtype and FP32:
        return x
    assert x.is_contiguous() and x.numel() % 8 == 0
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("mfnet_cast", x.data_ptr(), _DT[x.dtype], y.data_ptr(), _DT[dtype], x.numel(), _stream(x))
    return y


# The residual-stream gradient of a ViT block is written by the LayerNorm backward kernel in fp32 (for the residual add
# of the next backward) and, by the same launch, in bf16 (the GEMM operand of the next backward). Autograd carries only
# the fp32 tensor, so the bf16 copy waits here for the next backward to claim it: ONE slot, holding the fp32 tensor too
# (its memory cannot be reused while the slot is set), claimed by address, shape and version counter.
_twin = [None]


def remember_bf16_twin(d, d_bf):
    _twin[0] = (d, d_bf, d._version)


def bf16_of(d2):
    """bf16 copy of the fp32 gradient d2: the stored twin if d2 is that very tensor, else a cast launch."""
    t, _twin[0] = _twin[0], None
    if FP32:
        return d2
    if t is not None and t[0].data_ptr() == d2.data_ptr() and t[0].numel() == d2.numel() and t[0]._version == t[2] \
            and d2.is_contiguous():
        return t[1].view(d2.shape)
    return cast(d2, torch.bfloat16)


def gelu_fwd(x):
    """erf GELU on an fp32 tensor (standalone MLPBlock forwards; the blocks fuse it into the GEMM epilogue)."""
    _chk(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    y = torch.empty_like(x)
    call("mfnet_gelu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), _stream(x))
    return y


def gelu_bwd(x, dy):
    _chk(x, dy)
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == torch.float32 and dy.dtype == torch.float32
    dx = torch.empty_like(x)
    call("mfnet_gelu_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream(x))
    return dx


def im2col_patch(rgb, dsm, patch=16):
    """rgb [B,3,H,W] fp32, dsm [B,H,W] (one plane, replicated) or [B,3,H,W] fp32 -> [2*B*(H/P)*(W/P), 3*P*P] bf16 (rgb rows first)."""
    _chk(rgb, dsm)
    assert rgb.is_contiguous() and dsm.is_contiguous() and rgb.dtype == torch.float32 and dsm.dtype == torch.float32
    B, _, H, W = rgb.shape
    ych = 1 if dsm.dim() == 3 else int(dsm.shape[1])
    if tuple(dsm.shape) != ((B, H, W) if dsm.dim() == 3 else (B, ych, H, W)) or rgb.shape[1] != 3 or ych not in (1, 3):
        raise ValueError(f"im2col_patch: rgb {tuple(rgb.shape)} and second modality {tuple(dsm.shape)} do not match")
    if FP32:
        cols = torch.empty((2 * B * (H // patch) * (W // patch), 3 * patch * patch), dtype=torch.float32, device=rgb.device)
        call("mfnet_im2col_patch_f32", rgb.data_ptr(), dsm.data_ptr(), ych, cols.data_ptr(), B, H, W, patch, _stream(rgb))
        return cols
    cols = torch.empty((2 * B * (H // patch) * (W // patch), 3 * patch * patch), dtype=torch.bfloat16, device=rgb.device)
    tcall("mfnet_im2col_patch", (rgb.numel() + dsm.numel()) * 4 + cols.numel() * 2, 0, rgb.data_ptr(), dsm.data_ptr(), ych, cols.data_ptr(), B, H, W, patch, _stream(rgb))
    return cols


def im2col3x3(x, G, H, W):
    _chk(x)
    C = x.shape[-1]
    if x.dtype == torch.float32:
        assert x.is_contiguous()
        cols = torch.empty((G * H * W, 9 * C), dtype=torch.float32, device=x.device)
        call("mfnet_im2col3x3_f32", x.data_ptr(), cols.data_ptr(), G, H, W, C, _stream(x))
        return cols
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


def conv3x3(x, w, G, H, W, out_dtype=torch.float32):
    """3x3 / pad 1 convolution as an implicit GEMM: x [G*H*W, C] bf16 channels-last rows (C % 64 == 0), w [N, 9*C] bf16 with
    k = tap*C + c -> [G*H*W, N]. The data gradient is the same call on dy with the flipped, transposed weights."""
    _chk(x, w)
    if x.dtype == torch.float32:             # fp32 training: the column tensor and the fp32 GEMM (any C % 4 == 0)
        return gemm_f32(im2col3x3(x.contiguous(), G, H, W), w)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    C, N = x.shape[1], w.shape[0]
    assert x.shape[0] == G * H * W and w.shape[1] == 9 * C
    out = torch.empty((x.shape[0], N), dtype=out_dtype, device=x.device)
    call("mfnet_conv3x3_bf16", x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), G, H, W, C, N, out.data_ptr(), _dt(out), N, _stream(x))
    return out


def conv3x3_wgrad(dy, x, G, H, W, want_bias=False):
    """weight gradient of conv3x3: [N, 9*C] fp32 (k = tap*C + c) as a SplitSum over the row splits (a plain tensor when there is
    one split); dy [G*H*W, N] bf16, x [G*H*W, C] bf16, C % 64 == 0. want_bias: (dW, db) with db = column sums of dy."""
    _chk(dy, x)
    if dy.dtype == torch.float32:
        dW = gemm_f32(_t32(dy), _t32(im2col3x3(x.contiguous(), G, H, W)))
        return (dW, SplitSum(colsum(dy.contiguous(), 1, dy.shape[0])[0])) if want_bias else dW
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.stride(-1) == 1 and x.stride(-1) == 1
    M, N = dy.shape
    C = x.shape[1]
    assert x.shape[0] == M == G * H * W
    K = 9 * C
    splits = int(_lib.lib().mfnet_gemm_tn_splits(M, N, K))
    part = torch.empty((splits, N, K), dtype=torch.float32, device=dy.device)
    bpart = torch.empty((splits, N), dtype=torch.float32, device=dy.device) if want_bias else None
    if TN_QUEUE is not None:
        _tn_enqueue((dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K, part.data_ptr(), K, splits, 1,
                     bpart.data_ptr() if want_bias else 0, H, W, C, 0), (dy, x, part, bpart))
    else:
        call("mfnet_gemm_tn_conv3x3", dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), G, H, W, C, N, part.data_ptr(), K, splits,
             bpart.data_ptr() if want_bias else None, _stream(dy))
    dW = part[0] if splits == 1 else SplitSum(part)
    return (dW, SplitSum(bpart)) if want_bias else dW


def lora_mix_fwd(a, wx, wy):
    _chk(a, wx, wy)
    M, r = a.shape
    U = torch.empty((M, 2 * r), dtype=torch.float32, device=a.device)
    call("mfnet_lora_mix_fwd", a.data_ptr(), wx.data_ptr(), wy.data_ptr(), U.data_ptr(), M, r, _stream(a))
    return U


def lora_mix_bwd(P, a, wx, wy):
    """P: [M, 2r] or a stack of partial sums [nparts, M, 2r] (the column strips of lora_side) -> (da [M, r], dw [2])."""
    _chk(P, a, wx, wy)
    M, r = a.shape
    assert P.is_contiguous() and P.shape[-2:] == (M, 2 * r)
    nparts = P.shape[0] if P.dim() == 3 else 1
    da = torch.empty_like(a)
    dw = zeros(2, a.device)
    call("mfnet_lora_mix_bwd", P.data_ptr(), nparts, M * 2 * r, a.data_ptr(), wx.data_ptr(), wy.data_ptr(), da.data_ptr(),
         dw.data_ptr(), M, r, _stream(a))
    return da, dw


def lora_side(y, U=None, Vt=None, n_groups=1, scale=1.0):
    """One pass over y [M, C] bf16 for both rank-r products of a LoRA backward:
    dV = SplitSum of [G, C, R] partial tiles of y^T U (U fp32 [M, R], R = 4 or 8; rows of group g only), and
    P_parts [strips, M, 8] fp32 whose sum over dim 0 is y . Vt[g]^T (Vt fp32 [RP, C] or [G, RP, C], RP <= 8; columns >= RP are 0).
    Either product may be left out (None)."""
    _chk(y, U, Vt)
    y2, M, C, ldy = _rows2d(y)
    assert y.dtype == torch.bfloat16 and M % n_groups == 0
    Mg = M // n_groups
    dv = pp = None
    R = 8
    if U is not None:
        R = U.shape[1]
        assert U.dtype == torch.float32 and U.stride(1) == 1 and U.shape[0] == M
        splits = int(_lib.lib().mfnet_lora_side_splits(Mg, C))
        dv = torch.empty((n_groups, splits, C, R), dtype=torch.float32, device=y.device)
    RP, ldvt, vgs = 0, 0, 0
    if Vt is not None:
        assert Vt.dtype == torch.float32 and Vt.is_contiguous() and Vt.shape[-1] == C
        RP, ldvt = Vt.shape[-2], C
        vgs = Vt.stride(0) if Vt.dim() == 3 else 0
        assert Vt.dim() == 2 or Vt.shape[0] == n_groups
        pp = torch.empty(((C + 255) // 256, M, 8), dtype=torch.float32, device=y.device)
    if TN_QUEUE is not None and Vt is None and R in (4, 8):
        # the column product alone (dA = x^T . da): nothing but the gradient delivery reads it - collected like the wgrad GEMMs
        _tn_enqueue((y2.data_ptr(), ldy, Mg, C, n_groups, U.data_ptr(), U.stride(0), dv.data_ptr()), (y2, U, dv), kind=("ls", R))
    else:
        tcall("mfnet_lora_side", M * C * 2 + (pp.numel() * 4 if pp is not None else 0) + (dv.numel() * 4 if dv is not None else 0), 0,
              y2.data_ptr(), ldy, Mg, C, n_groups, _p(U), U.stride(0) if U is not None else 0, R, _p(dv),
             _p(Vt), ldvt, vgs, RP, _p(pp), _stream(y))
    if dv is not None:
        dv = SplitSum(dv.transpose(0, 1), scale)               # [splits, G, C, R], summed (x scale) by the gradient delivery
        if n_groups == 1:
            dv = dv[0]
    return dv, pp


def reduce_parts(parts, scale=1.0):
    """scale * sum over dim 0 of a contiguous fp32 stack (e.g. the P partials of lora_side)."""
    _chk(parts)
    assert parts.is_contiguous() and parts.dtype == torch.float32
    out = torch.empty(parts.shape[1:], dtype=torch.float32, device=parts.device)
    assert out.numel() % 4 == 0
    call("mfnet_reduce_splits", parts.data_ptr(), out.data_ptr(), out.numel(), parts.shape[0], float(scale), 0, _stream(parts))
    return out


# ------------------------------------------------------------------------------------------ attention
def attn_fwd(qkv, qkv_bias, tbl, G, Sh, Sw, heads, ws, Lh, Lw, scale):
    """qkv [G*Sh*Sw, 3D] bf16 -> (out [G*Sh*Sw, D] bf16, lse [G, heads, Sh*Sw] fp32)."""
    _chk(qkv, qkv_bias, tbl)
    D = qkv.shape[1] // 3
    hd = D // heads
    assert qkv.is_contiguous() and tbl.is_contiguous() and tbl.dtype == torch.bfloat16
    out = torch.empty((qkv.shape[0], D), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((G, heads, Sh * Sw), dtype=torch.float32, device=qkv.device)
    nk = (ws * ws if ws > 0 else Sh * Sw)
    kk = (2 * ws if ws > 0 else Sh + Sw)
    fl = 4.0 * G * heads * Sh * Sw * nk * hd + 2.0 * G * heads * Sh * Sw * kk * hd      # QK^T + PV over real queries x window keys, + the rel-pos products
    tcall("mfnet_attn_fwd", qkv.numel() * 2 + out.numel() * 2, fl, qkv.data_ptr(), qkv_bias.data_ptr(), tbl.data_ptr(), out.data_ptr(), lse.data_ptr(),
         G, Sh, Sw, heads, hd, ws, Lh, Lw, tbl.shape[0], float(scale), _stream(qkv))
    return out, lse


def attn_bwd(qkv, qkv_bias, tbl, tblT, out, dout, lse, G, Sh, Sw, heads, ws, Lh, Lw, scale):
    _chk(qkv, qkv_bias, tbl, tblT, out, dout, lse)
    D = qkv.shape[1] // 3
    hd = D // heads
    assert dout.is_contiguous() and dout.dtype == torch.bfloat16
    dqkv = torch.empty_like(qkv)
    nws = int(_lib.lib().mfnet_attn_bwd_workspace_floats(G, Sh, Sw, heads, ws))
    wsp = torch.empty(nws, dtype=torch.float32, device=qkv.device) if nws else None
    nk = (ws * ws if ws > 0 else Sh * Sw)
    kk = (2 * ws if ws > 0 else Sh + Sw)
    fl = 2.0 * (4.0 * G * heads * Sh * Sw * nk * hd + 2.0 * G * heads * Sh * Sw * kk * hd)      # backward = 2 x forward
    tcall("mfnet_attn_bwd", qkv.numel() * 2 * 2 + out.numel() * 2 * 2, fl, qkv.data_ptr(), qkv_bias.data_ptr(), tbl.data_ptr(), tblT.data_ptr(), out.data_ptr(),
         dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), G, Sh, Sw, heads, hd, ws, Lh, Lw, tbl.shape[0],
         float(scale), _p(wsp), nws, _stream(qkv))
    return dqkv


def attn_f32(qkv, qkv_bias, th, tw, G, Sh, Sw, heads, ws, scale):
    """SAM ViT attention in fp32 (csrc/fp32.hip): qkv [G*Sh*Sw, 3D] fp32, th / tw the resized fp32 rel-pos tables -> out [G*Sh*Sw, D]."""
    _chk(qkv, qkv_bias, th, tw)
    assert qkv.is_contiguous() and qkv.dtype == torch.float32 and th.is_contiguous() and tw.is_contiguous()
    D3 = qkv.shape[1]
    out = torch.empty((qkv.shape[0], D3 // 3), dtype=torch.float32, device=qkv.device)
    call("mfnet_attn_f32", qkv.data_ptr(), qkv_bias.data_ptr(), th.data_ptr(), tw.data_ptr(), out.data_ptr(), G, Sh, Sw, heads,
         D3 // (3 * heads), ws, float(scale), _stream(qkv))
    return out


def attn_f32_bwd(qkv, qkv_bias, th, tw, out, dout, G, Sh, Sw, heads, ws, scale):
    """dqkv of attn_f32 (tables and qkv bias frozen)."""
    _chk(qkv, qkv_bias, th, tw, out, dout)
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and qkv.dtype == out.dtype == dout.dtype == torch.float32
    D3 = qkv.shape[1]
    dqkv = torch.empty_like(qkv)
    nws = int(_lib.lib().mfnet_attn_f32_bwd_workspace_floats(G, Sh, Sw, heads, ws))
    wsp = torch.empty(nws, dtype=torch.float32, device=qkv.device)
    call("mfnet_attn_f32_bwd", qkv.data_ptr(), qkv_bias.data_ptr(), th.data_ptr(), tw.data_ptr(), out.data_ptr(), dout.data_ptr(),
         dqkv.data_ptr(), G, Sh, Sw, heads, D3 // (3 * heads), ws, float(scale), wsp.data_ptr(), nws, _stream(qkv))
    return dqkv


# ------------------------------------------------------------------------------------------ train / infer glue
def ce_loss_fwd(logits, target, weight=None, ignore_index=-100, want_pred=False):
    _chk(logits, target, weight)
    assert logits.is_contiguous() and logits.dtype == torch.float32 and target.dtype == torch.int64 and target.is_contiguous()
    B, C = logits.shape[:2]
    HW = logits.numel() // (B * C)
    acc = torch.zeros(4, dtype=torch.float32, device=logits.device)   # sum w*nll, sum w, #correct, #valid
    pred = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.uint8, device=logits.device) if want_pred else None
    tcall("mfnet_ce_loss_fwd", logits.numel() * 4 + target.numel() * 8, 0, logits.data_ptr(), target.data_ptr(), _p(weight), acc.data_ptr(), _p(pred),
         B, C, HW, ignore_index, _stream(logits))
    return acc, pred


def ce_loss_bwd(logits, target, acc, weight=None, ignore_index=-100, gscale=1.0):
    B, C = logits.shape[:2]
    HW = logits.numel() // (B * C)
    d = torch.empty_like(logits)
    tcall("mfnet_ce_loss_bwd", logits.numel() * 8 + target.numel() * 8, 0, logits.data_ptr(), target.data_ptr(), _p(weight), acc.data_ptr(), d.data_ptr(),
         float(gscale), B, C, HW, ignore_index, _stream(logits))
    return d


def sgd_flat(p, g, buf, lr, momentum, wd, gscale=1.0):
    _chk(p, g, buf)
    assert p.is_contiguous() and g.is_contiguous() and p.numel() == g.numel()
    tcall("mfnet_sgd_flat", p.numel() * 4 * (4 if buf is None else 5) - (p.numel() * 4 if buf is None else 0), 0, p.data_ptr(), g.data_ptr(), _p(buf), p.numel(), float(lr), float(momentum), float(wd),
         float(gscale), _stream(p))


def tile_accumulate(pred, logits, coords, atomic=True):
    """pred [C,H,W] += logits [n,C,h,w] placed at coords [n,2] int32 (row, col). pred fp32 (logits fp32) or float64 (logits
    fp32 or float64): the float64 accumulator is the reference's own (evaluator.py:134) - exact sums, order-independent."""
    _chk(pred, logits, coords)
    assert pred.is_contiguous() and logits.is_contiguous() and coords.dtype == torch.int32 and coords.is_contiguous()
    C, H, W = pred.shape
    n, C2, h, w = logits.shape
    assert C2 == C
    if pred.dtype == torch.float64:
        assert logits.dtype in (torch.float32, torch.float64)
        call("mfnet_tile_accumulate_f64", pred.data_ptr(), logits.data_ptr(), 1 if logits.dtype == torch.float64 else 0, coords.data_ptr(),
             n, C, H, W, h, w, 1 if atomic else 0, _stream(pred))
        return
    assert pred.dtype == torch.float32 and logits.dtype == torch.float32
    call("mfnet_tile_accumulate", pred.data_ptr(), logits.data_ptr(), coords.data_ptr(), n, C, H, W, h, w,
         1 if atomic else 0, _stream(pred))


def argmax_chw(pred):
    _chk(pred)
    assert pred.is_contiguous() and pred.dtype in (torch.float32, torch.float64)
    C = pred.shape[0]
    out = torch.empty(pred.shape[1:], dtype=torch.uint8, device=pred.device)
    call("mfnet_argmax_chw_f64" if pred.dtype == torch.float64 else "mfnet_argmax_chw", pred.data_ptr(), out.data_ptr(), C, out.numel(), _stream(pred))
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
    dw = zeros((16, 2), dh.device)                       # 16 slots of (d wx, d wy); summed by the gradient delivery
    if FP32:
        dada = torch.empty((M, D), dtype=torch.float32, device=dh.device)
        call("mfnet_adapter_mix_bwd_f32", dh.data_ptr(), ada.data_ptr(), wx.data_ptr(), wy.data_ptr(), dada.data_ptr(), dw.data_ptr(), M, D, _stream(dh))
        return dada, SplitSum(dw)
    dada = torch.empty((M, D), dtype=torch.bfloat16, device=dh.device)
    call("mfnet_adapter_mix_bwd", dh.data_ptr(), ada.data_ptr(), wx.data_ptr(), wy.data_ptr(), dada.data_ptr(),
         dw.data_ptr(), None, M, D, _stream(dh))
    return dada, SplitSum(dw)


def bias_grad(dy, out=None):
    _chk(dy)
    dy2, M, N, ld = _rows2d(dy)
    if out is None and ld == N and N % 4 == 0 and N <= 1024 and dy2.is_contiguous():
        # the deterministic column reduction (16-byte loads for bf16), no float atomics
        return colsum(dy2, 1, M)[0].view(N)
    if out is None:
        out = zeros(N, dy.device)
    call("mfnet_bias_grad", dy2.data_ptr(), ld, out.data_ptr(), M, N, _stream(dy))
    return out


def add_bf16(a, b, out):
    _chk(a, b, out)
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
    if a.dtype == torch.float32:
        return add3(a, b, out=out)
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


# Weight-gradient products whose result is only handed to the gradient delivery (lazy SplitSums) need not run where the
# backward pass meets them: while a list is installed here (bank.ParamBank with deferred delivery does that) they are collected
# - operands kept alive by the entries - and tn_flush() runs them all in one mfnet_gemm_tn_batch launch before the delivery's
# gather reads their partial tiles. None: every product is launched on the spot.
TN_QUEUE = None


def _tn_enqueue(desc, keep, kind="tn"):
    TN_QUEUE.append((kind, desc, keep))


def tn_flush():
    """launch the collected weight-gradient products - one mfnet_gemm_tn_batch (longest row ranges first: the long ones start
    early, the small ones fill the gaps) and one mfnet_lora_side_batch per LoRA rank; the queue stays installed"""
    q = TN_QUEUE
    if not q:
        return
    import ctypes
    entries = list(q)
    del q[:]
    tn = sorted((e for e in entries if e[0] == "tn"), key=lambda e: -(e[1][4] // max(1, e[1][9])))          # rows per split
    if tn:
        flat = [v for _, d, _ in tn for v in d]
        arr = (ctypes.c_int64 * len(flat))(*flat)
        call("mfnet_gemm_tn_batch", ctypes.cast(arr, ctypes.c_void_p), len(tn), _stream(tn[0][2][0]))
    dwq = [e for e in entries if e[0] == "dw"]
    if dwq:
        flat = [v for _, d, _ in dwq for v in d]
        arr = (ctypes.c_int64 * len(flat))(*flat)
        call("mfnet_dwconv_bwd_weight_batch", ctypes.cast(arr, ctypes.c_void_p), len(dwq), _stream(dwq[0][2][0]))
    for R in (4, 8):
        ls = [e for e in entries if e[0] == ("ls", R)]
        if ls:
            flat = [v for _, d, _ in ls for v in d]
            arr = (ctypes.c_int64 * len(flat))(*flat)
            call("mfnet_lora_side_batch", ctypes.cast(arr, ctypes.c_void_p), len(ls), R, _stream(ls[0][2][0]))


def gemm_tn_grouped(dy, x, n_groups, want_bias=False):
    """n_groups independent dW_g[n,k] = sum over the rows of group g (M / n_groups consecutive rows each) of dy[m,n] x[m,k]
    in one launch -> SplitSum of shape [G, N, K] (index it per group and hand the pieces to deliver). want_bias: also the
    column sums of dy per group (the bias gradients), from the same launch -> (dW, db) with db a SplitSum [G, N]."""
    _chk(dy, x)
    if dy.dtype == torch.float32:
        d2, M, N, ldy = _rows2d(dy)
        x2, M2, K, ldx = _rows2d(x)
        assert M == M2 and M % n_groups == 0 and x.dtype == torch.float32
        Mg = M // n_groups
        part = torch.empty((1, n_groups, N, K), dtype=torch.float32, device=dy.device)
        bpart = torch.empty((1, n_groups, N), dtype=torch.float32, device=dy.device) if want_bias else None
        for g in range(n_groups):
            dg = d2[g * Mg:(g + 1) * Mg]
            gemm_f32(_t32(dg), _t32(x2[g * Mg:(g + 1) * Mg]), out=part[0, g])
            if want_bias:
                bpart[0, g].copy_(colsum(dg.contiguous(), 1, Mg)[0].view(N))
        return (SplitSum(part), SplitSum(bpart)) if want_bias else SplitSum(part)
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.stride(-1) == 1 and x.stride(-1) == 1
    d2, M, N, ldy = _rows2d(dy)
    x2, M2, K, ldx = _rows2d(x)
    assert M == M2 and M % n_groups == 0
    Mg = M // n_groups
    splits = int(_lib.lib().mfnet_gemm_tn_splits(Mg, N, K * n_groups))
    part = torch.empty((splits, n_groups, N, K), dtype=torch.float32, device=dy.device)
    bpart = torch.empty((splits, n_groups, N), dtype=torch.float32, device=dy.device) if want_bias else None
    if TN_QUEUE is not None:
        _tn_enqueue((d2.data_ptr(), ldy, x2.data_ptr(), ldx, Mg, N, K, part.data_ptr(), K, splits, n_groups,
                     bpart.data_ptr() if want_bias else 0, 0, 0, 0, 0), (d2, x2, part, bpart))
    else:
        call("mfnet_gemm_tn_bf16_grouped", d2.data_ptr(), ldy, x2.data_ptr(), ldx, Mg, N, K, part.data_ptr(), K, splits, n_groups,
             bpart.data_ptr() if want_bias else None, _stream(dy))
    return (SplitSum(part), SplitSum(bpart)) if want_bias else SplitSum(part)


def gemm_tn(dy, x, lazy=False, want_bias=False):
    """dW[n,k] = sum_m dy[m,n] x[m,k]  (fp32 [N,K]); both bf16 row-major with M leading (views with a row
    stride are fine). The reduction runs over the rows (pixels / tokens): it is cut into splits whose partial
    tiles one small kernel sums - or, with lazy, the gradient delivery does (returns a SplitSum). want_bias: the launch
    also leaves the column sums of dy (the layer's bias gradient) -> (dW, db), db always a SplitSum over [splits, N]
    partial rows (view its .part to fold further, e.g. the 4 taps of a transposed conv)."""
    _chk(dy, x)
    if dy.dtype == torch.float32:
        d2, M, N, ldy = _rows2d(dy)
        x2, M2, K, ldx = _rows2d(x)
        assert M == M2 and x.dtype == torch.float32
        dW = gemm_f32(_t32(d2), _t32(x2))
        return (dW, SplitSum(colsum(d2.contiguous(), 1, M)[0])) if want_bias else dW
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.stride(-1) == 1 and x.stride(-1) == 1
    d2, M, N, ldy = _rows2d(dy)
    x2, M2, K, ldx = _rows2d(x)
    assert M == M2
    splits = int(_lib.lib().mfnet_gemm_tn_splits(M, N, K))
    part = torch.empty((splits, N, K), dtype=torch.float32, device=dy.device)
    bpart = torch.empty((splits, N), dtype=torch.float32, device=dy.device) if want_bias else None
    if TN_QUEUE is not None and lazy:
        _tn_enqueue((d2.data_ptr(), ldy, x2.data_ptr(), ldx, M, N, K, part.data_ptr(), K, splits, 1,
                     bpart.data_ptr() if want_bias else 0, 0, 0, 0, 0), (d2, x2, part, bpart))
    else:
        call("mfnet_gemm_tn_bf16", d2.data_ptr(), ldy, x2.data_ptr(), ldx, M, N, K, part.data_ptr(), K, splits,
             bpart.data_ptr() if want_bias else None, _stream(dy))
    if splits == 1:
        dW = part[0]
    elif lazy:
        dW = SplitSum(part)
    else:
        dW = torch.empty((N, K), dtype=torch.float32, device=dy.device)
        call("mfnet_reduce_splits", part.data_ptr(), dW.data_ptr(), dW.numel(), splits, 1.0, 0, _stream(part))
    return (dW, SplitSum(bpart)) if want_bias else dW


# ------------------------------------------------------------------------------------------ FPN / fusion / decoder
# All maps are channels-last [rows, C]; see csrc/decoder.hip.
ACT_NONE, ACT_RELU6, ACT_GELU = 0, 1, 2


def _dt(t):
    return _DT[t.dtype]


_TICKETS = {}


def _colreduce_ws(G, P, C, device):
    """(partial-slab workspace, zeroed tickets) of a deterministic column reduction. The tickets are one
    persistent zero buffer per device: every reduction leaves them zero again, and calls on one device are
    stream-ordered."""
    n = int(_lib.lib().mfnet_colreduce_workspace_floats(G, P, C))
    t = _TICKETS.get(device)
    if t is None or t.numel() < G:
        t = _TICKETS[device] = torch.zeros(max(1024, G), dtype=torch.int32, device=device)
    return torch.empty(max(n, 1), dtype=torch.float32, device=device), t


def colsum(x, G, P, want_sq=False):
    """x [G*P, C] -> (sum [G,C], sumsq [G,C] | None), fp32."""
    _chk(x)
    C = x.shape[-1]
    assert x.is_contiguous() and x.numel() == G * P * C
    s = zeros((2 if want_sq else 1, G, C), x.device)
    ws, tk = _colreduce_ws(G, P, C, x.device)
    tcall("mfnet_colsum", x.numel() * x.element_size(), 0, x.data_ptr(), _dt(x), s[0].data_ptr(), s[1].data_ptr() if want_sq else None, G, P, C,
         ws.data_ptr(), tk.data_ptr(), _stream(x))
    return s[0], (s[1] if want_sq else None)


def coldot(a, b, G, P):
    _chk(a, b)
    C = a.shape[-1]
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel() == G * P * C
    out = zeros((G, C), a.device)
    ws, tk = _colreduce_ws(G, P, C, a.device)
    tcall("mfnet_coldot", a.numel() * (a.element_size() + b.element_size()), 0, a.data_ptr(), _dt(a), b.data_ptr(), _dt(b), out.data_ptr(), G, P, C, ws.data_ptr(), tk.data_ptr(), _stream(a))
    return out


# Synchronised BatchNorm (strict 1-vs-N equivalence of data-parallel training; torch.nn.SyncBatchNorm's arithmetic):
# None = per-replica statistics. TrainEngine(sync_bn=True) sets it to (process_group, this rank's batch, the group's batch) - then the training-mode column sums
# (sum x, sum x^2, rows) and the backward sums (sum dy, sum dy*xhat, rows) are all-reduced over the group before they are used.
# The collectives sit between kernels of the step, so this mode runs eagerly (no HIP-graph replay).
SYNC_BN = None


def _sync_sums(parts, rows):
    """all-reduce the per-channel sums over the SyncBN group -> (summed parts, total rows). The total row count is host
    arithmetic - rows * (group batch / this rank's batch), both exchanged ONCE per step by the engine as integers (SYNC_BN[1:]) -
    not a float32 element of the reduced vector read back per layer: that cost two host syncs per BatchNorm and is exact only up
    to 2^24 rows (16 tiles of 1024 x 1024 per rank reach it at the decoder's full-resolution maps)."""
    import torch.distributed as dist
    pg, b_mine, b_total = SYNC_BN
    flat = torch.cat([p.reshape(-1) for p in parts])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=pg)
    out, o = [], 0
    for p in parts:
        out.append(flat[o:o + p.numel()].view(p.shape))
        o += p.numel()
    assert (rows * b_total) % b_mine == 0, "SyncBN: a map's row count must be a multiple of the rank's batch size"
    return out, rows * b_total // b_mine


WEIGHTS_EPOCH = [0]      # bumped (bank.weights_changed) by code that rewrites parameters / buffers without a torch in-place op
_BN_EVAL = None          # the active bank.EvalBank's store of eval-mode BatchNorm mean / rstd (None: no store)


def bn_eval_check(ent):
    """stored eval statistics [st, sig, run_mean, run_var, eps] re-derived in place if the running statistics changed"""
    st, sig, run_mean, run_var, eps = ent
    cur = (WEIGHTS_EPOCH[0], run_mean._version, run_var._version)
    if cur != sig:
        C = run_mean.shape[0]
        call("mfnet_bn_finalize", None, None, st[0].data_ptr(), st[1].data_ptr(), run_mean.data_ptr(), run_var.data_ptr(), None,
             C, 1, eps, 0.1, 0, _stream(st))
        ent[1] = cur


def bn_forward(x, gamma, beta, run_mean, run_var, nbt, training, eps=1e-5, momentum=0.1, act=ACT_NONE, chan_scale=None,
               rows_per_image=0, out_dtype=torch.float32, out2_dtype=None):
    """BatchNorm2d on a channels-last [rows, C] map. -> (y, y2 | None, mean, rstd)."""
    _chk(x, gamma, beta)
    out_dtype, out2_dtype = _odt(out_dtype), (_odt(out2_dtype) if out2_dtype is not None else None)
    rows, C = x.shape
    assert x.is_contiguous()
    st = torch.empty((2, C), dtype=torch.float32, device=x.device)
    y = torch.empty((rows, C), dtype=out_dtype, device=x.device)
    y2 = torch.empty((rows, C), dtype=out2_dtype, device=x.device) if out2_dtype is not None else None
    if training and run_mean is not None:
        WEIGHTS_EPOCH[0] += 1            # the kernels below rewrite the running statistics (no version bump)
    n_stat = rows
    if training and SYNC_BN is not None:
        s, sq = colsum(x, 1, rows, want_sq=True)
        (s, sq), n_stat = _sync_sums([s, sq], rows)
        call("mfnet_bn_finalize", s.data_ptr(), sq.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), _p(run_mean), _p(run_var),
             _p(nbt), C, n_stat, float(eps), float(momentum), 1, _stream(x))
    elif training and C <= 1024:
        s, sq = colsum(x, 1, rows, want_sq=True)
        tcall("mfnet_bn_train_apply", rows * C * (x.element_size() + y.element_size() + (y2.element_size() if y2 is not None else 0)), 0,
              x.data_ptr(), _dt(x), s.data_ptr(), sq.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), _p(run_mean),
             _p(run_var), _p(nbt), float(eps), float(momentum), gamma.data_ptr(), beta.data_ptr(), act, _p(chan_scale), rows_per_image,
             y.data_ptr(), _dt(y), _p(y2), _dt(y2) if y2 is not None else 0, rows, C, _stream(x))
        return y, y2, st[0], st[1]
    if training and SYNC_BN is not None:
        pass
    elif training:
        s, sq = colsum(x, 1, rows, want_sq=True)
        call("mfnet_bn_finalize", s.data_ptr(), sq.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), _p(run_mean), _p(run_var),
             _p(nbt), C, rows, float(eps), float(momentum), 1, _stream(x))
    else:
        # eval: mean / rstd follow from the running statistics alone - inside a predictor's EvalBank they are kept between calls
        # (validated like bank.derived's operands: version counters + WEIGHTS_EPOCH), one launch less per BatchNorm and forward
        key = (run_mean.data_ptr(), run_var.data_ptr(), float(eps), C)
        store = _BN_EVAL if (_BN_EVAL is not None and not torch.is_grad_enabled()) else None
        ent = store.get(key) if store is not None else None
        capturing = torch.cuda.is_current_stream_capturing()
        if ent is not None:
            if not capturing:
                bn_eval_check(ent)
            st = ent[0]
        else:
            call("mfnet_bn_finalize", None, None, st[0].data_ptr(), st[1].data_ptr(), run_mean.data_ptr(), run_var.data_ptr(), None,
                 C, rows, float(eps), float(momentum), 0, _stream(x))
            if store is not None and not capturing:
                store[key] = [st, (WEIGHTS_EPOCH[0], run_mean._version, run_var._version), run_mean, run_var, float(eps)]
    tcall("mfnet_bn_apply", rows * C * (x.element_size() + y.element_size() + (y2.element_size() if y2 is not None else 0)), 0,
          x.data_ptr(), _dt(x), st[0].data_ptr(), st[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), act,
         _p(chan_scale), rows_per_image, y.data_ptr(), _dt(y), _p(y2), _dt(y2) if y2 is not None else 0, rows, C, _stream(x))
    return y, y2, st[0], st[1]


def bn_backward(dy, x, mean, rstd, gamma, beta, training, act=ACT_NONE, chan_scale=None, rows_per_image=0, dx_dtype=torch.float32):
    """-> (dx, dgamma, dbeta)."""
    _chk(dy, x)
    dx_dtype = _odt(dx_dtype)
    rows, C = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dy.shape == x.shape
    sums = zeros((2, C), x.device)
    ws, tk = _colreduce_ws(1, rows, C, x.device)
    tcall("mfnet_bn_bwd_reduce", rows * C * (dy.element_size() + x.element_size()), 0, dy.data_ptr(), _dt(dy), x.data_ptr(), _dt(x), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
         beta.data_ptr(), act, _p(chan_scale), rows_per_image, sums[0].data_ptr(), sums[1].data_ptr(), rows, C,
         ws.data_ptr(), tk.data_ptr(), _stream(x))
    dx = torch.empty((rows, C), dtype=dx_dtype, device=x.device)
    use = sums
    if training and SYNC_BN is not None:
        # dx takes the group-wide means of dy and dy*xhat (the kernel divides by its own row count: pre-scale by rows / total);
        # dgamma / dbeta stay this rank's sums - the gradient exchange adds the ranks
        (g,), n_tot = _sync_sums([sums.clone()], rows)
        use = g * (float(rows) / float(n_tot))
    tcall("mfnet_bn_bwd_apply", rows * C * (dy.element_size() + x.element_size() + dx.element_size()), 0, dy.data_ptr(), _dt(dy), x.data_ptr(), _dt(x), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
         beta.data_ptr(), act, _p(chan_scale), rows_per_image, use[0].data_ptr(), use[1].data_ptr(), 1 if training else 0,
         dx.data_ptr(), _dt(dx), rows, C, _stream(x))
    return dx, sums[1], sums[0]


def act(x, kind, out_dtype=None):
    _chk(x)
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=_odt(out_dtype or x.dtype), device=x.device)
    call("mfnet_act", x.data_ptr(), _dt(x), y.data_ptr(), _dt(y), kind, x.numel(), _stream(x))
    return y


def _relu6(x):
    return act(x, ACT_RELU6)


def act_bwd(dy, ref, kind, out_dtype=torch.bfloat16):
    _chk(dy, ref)
    assert dy.is_contiguous() and ref.is_contiguous() and dy.shape == ref.shape
    dx = torch.empty(dy.shape, dtype=_odt(out_dtype), device=dy.device)
    call("mfnet_act_bwd", dy.data_ptr(), _dt(dy), ref.data_ptr(), _dt(ref), dx.data_ptr(), _dt(dx), kind, dy.numel(), _stream(dy))
    return dx


def dwconv_fwd(x, w, bias, B, H, W, k, ext):
    _chk(x, w, bias)
    C = x.shape[-1]
    assert x.is_contiguous() and x.dtype == torch.float32 and w.is_contiguous() and w.numel() == C * k * k
    y = torch.empty_like(x)
    tcall("mfnet_dwconv_fwd", x.numel() * 8, 2.0 * x.numel() * k * k, x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), B, H, W, C, k, 1 if ext else 0, _stream(x))
    return y


def dwconv_bwd(dy, x, w, B, H, W, k, ext, want_bias=False):
    """-> (dx fp32, dw like w, dbias | None); dw / dbias are SplitSums (materialize() for a tensor)."""
    _chk(dy, x, w)
    C = x.shape[-1]
    assert dy.is_contiguous() and x.is_contiguous() and x.dtype == torch.float32
    dx = torch.empty_like(x)
    tcall("mfnet_dwconv_bwd_data", dy.numel() * dy.element_size() + x.numel() * 4, 2.0 * x.numel() * k * k, dy.data_ptr(), _dt(dy), w.data_ptr(), dx.data_ptr(), B, H, W, C, k, 1 if ext else 0, _stream(x))
    slots = 32                                           # copies of dw / db the workgroups spread their atomics over
    dw = zeros((slots,) + tuple(w.shape), x.device)
    db = zeros((slots, C), x.device) if want_bias else None
    if TN_QUEUE is not None:
        _tn_enqueue((dy.data_ptr(), _dt(dy), x.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else 0, B, H, W, C, k, 1 if ext else 0, slots),
                    (dy, x, dw, db), kind="dw")
    else:
        call("mfnet_dwconv_bwd_weight", dy.data_ptr(), _dt(dy), x.data_ptr(), dw.data_ptr(), _p(db), B, H, W, C, k, 1 if ext else 0, slots,
             _stream(x))
    return dx, SplitSum(dw), (SplitSum(db) if want_bias else None)       # summed over the slots by the gradient delivery


def gla_attn_fwd(qkv, table, B, H, W, heads, scale):
    _chk(qkv, table)
    assert qkv.is_contiguous() and qkv.dtype in (torch.bfloat16, torch.float32) and table.is_contiguous() and table.dtype == torch.float32
    out = torch.empty((qkv.shape[0], heads * 8), dtype=torch.float32, device=qkv.device)
    call("mfnet_gla_attn_fwd_f32" if qkv.dtype == torch.float32 else "mfnet_gla_attn_fwd", qkv.data_ptr(), table.data_ptr(), out.data_ptr(), B, H, W, heads, qkv.shape[1] // (3 * heads), 8, float(scale), _stream(qkv))
    return out


def gla_attn_bwd(qkv, table, dout, B, H, W, heads, scale):
    _chk(qkv, table, dout)
    assert dout.is_contiguous() and dout.dtype == torch.float32
    dqkv = torch.empty_like(qkv)
    dtable = zeros(table.shape, table.device)
    call("mfnet_gla_attn_bwd_f32" if qkv.dtype == torch.float32 else "mfnet_gla_attn_bwd", qkv.data_ptr(), table.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dtable.data_ptr(), B, H, W, heads,
         qkv.shape[1] // (3 * heads), 8, float(scale), _stream(qkv))
    return dqkv, dtable


def gla_pool_fwd(o, l1, l2, B, H, W):
    _chk(o, l1, l2)
    out = torch.empty_like(o)
    call("mfnet_gla_pool_fwd", o.data_ptr(), l1.data_ptr(), l2.data_ptr(), out.data_ptr(), B, H, W, o.shape[-1], _stream(o))
    return out


def gla_pool_bwd(ds, B, H, W):
    _chk(ds)
    assert ds.is_contiguous() and ds.dtype == torch.float32
    d_o = torch.empty_like(ds)
    call("mfnet_gla_pool_bwd", ds.data_ptr(), d_o.data_ptr(), B, H, W, ds.shape[-1], _stream(ds))
    return d_o


def upfuse_fwd(x, skip, perm, w, eps, B, H, W, out_dtype=torch.bfloat16):
    """x [B*(H/2)*(W/2), C] fp32, skip [B*H*W, C] (row order `perm`) -> [B*H*W, C]."""
    _chk(x, skip, w)
    C = x.shape[-1]
    assert x.is_contiguous() and x.dtype == torch.float32 and skip.is_contiguous() and skip.shape == (B * H * W, C)
    out = torch.empty((B * H * W, C), dtype=_odt(out_dtype), device=x.device)
    tcall("mfnet_upfuse_fwd", x.numel() * 4 + skip.numel() * skip.element_size() + out.numel() * out.element_size(), 0, x.data_ptr(), skip.data_ptr(), _dt(skip), perm, w.data_ptr(), float(eps), out.data_ptr(), _dt(out), B, H, W, C, _stream(x))
    return out


def upfuse_bwd(df, x, skip, perm, w, eps, B, H, W, dskip_dtype=torch.bfloat16):
    """-> (dx fp32, dskip, dw [2])."""
    _chk(df, x, skip, w)
    C = x.shape[-1]
    assert df.is_contiguous() and df.dtype == torch.float32
    dx = torch.empty_like(x)
    dskip = torch.empty(skip.shape, dtype=_odt(dskip_dtype), device=x.device)
    dw = zeros(2, x.device)
    sums = torch.empty(64, dtype=torch.float32, device=x.device)          # 32 slots x 2, zeroed by the entry point
    tcall("mfnet_upfuse_bwd", df.numel() * 4 + x.numel() * 8 + skip.numel() * skip.element_size() + dskip.numel() * dskip.element_size(), 0, df.data_ptr(), x.data_ptr(), skip.data_ptr(), _dt(skip), perm, w.data_ptr(), float(eps), dx.data_ptr(),
         dskip.data_ptr(), _dt(dskip), dw.data_ptr(), sums.data_ptr(), B, H, W, C, _stream(x))
    return dx, dskip, dw


def up_final_fwd(x, B, h, w, ncls, s):
    _chk(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    out = torch.empty((B, ncls, h * s, w * s), dtype=torch.float32, device=x.device)
    tcall("mfnet_up_final_fwd", B * h * w * ncls * 4 + out.numel() * 4, 0, x.data_ptr(), x.shape[-1], out.data_ptr(), B, h, w, ncls, s, _stream(x))
    return out


def up_final_bwd(dlogits, ldx, B, h, w, ncls, s):
    _chk(dlogits)
    assert dlogits.is_contiguous() and dlogits.dtype == torch.float32
    if FP32:
        dx = torch.empty((B * h * w, ldx), dtype=torch.float32, device=dlogits.device)
        call("mfnet_up_final_bwd_f32", dlogits.data_ptr(), dx.data_ptr(), ldx, B, h, w, ncls, s, _stream(dlogits))
        return dx
    dx = torch.empty((B * h * w, ldx), dtype=torch.bfloat16, device=dlogits.device)
    tcall("mfnet_up_final_bwd", dlogits.numel() * 4 + B * h * w * ncls * 2, 0, dlogits.data_ptr(), dx.data_ptr(), ldx, B, h, w, ncls, s, _stream(dlogits))
    return dx


def maxpool2_fwd(x, B, H, W):
    _chk(x)
    C = x.shape[-1]
    assert x.is_contiguous() and x.dtype == torch.float32
    y = torch.empty((B * (H // 2) * (W // 2), C), dtype=torch.float32, device=x.device)
    arg = torch.empty((B * (H // 2) * (W // 2), C), dtype=torch.uint8, device=x.device)
    call("mfnet_maxpool2_fwd", x.data_ptr(), y.data_ptr(), arg.data_ptr(), B, H, W, C, _stream(x))
    return y, arg


def maxpool2_bwd(dy, arg, dx, B, H, W, accumulate):
    _chk(dy, arg, dx)
    assert dy.is_contiguous() and dx.is_contiguous() and dx.dtype == torch.float32
    call("mfnet_maxpool2_bwd", dy.data_ptr(), _dt(dy), arg.data_ptr(), dx.data_ptr(), B, H, W, dx.shape[-1], 1 if accumulate else 0, _stream(dx))
    return dx


def se_gate_fwd(pooled_sum, P, W1, b1, W2, b2, relu6):
    _chk(pooled_sum, W1, W2)
    B, C = pooled_sum.shape
    Cr = W1.shape[0]
    z = torch.empty((B, Cr), dtype=torch.float32, device=W1.device)
    g = torch.empty((B, C), dtype=torch.float32, device=W1.device)
    call("mfnet_se_gate_fwd", pooled_sum.data_ptr(), P, W1.data_ptr(), _p(b1), W2.data_ptr(), _p(b2), 1 if relu6 else 0, z.data_ptr(),
         g.data_ptr(), B, C, Cr, _stream(W1))
    return z, g


def se_gate_bwd(dg, pooled_sum, P, W1, W2, relu6, z, g, has_bias):
    """-> (dpool [B,C] (already / P), dW1, db1|None, dW2, db2|None)."""
    B, C = pooled_sum.shape
    Cr = W1.shape[0]
    dev = W1.device
    dpool = torch.empty((B, C), dtype=torch.float32, device=dev)
    dW1 = zeros(W1.shape, dev)
    dW2 = zeros(W2.shape, dev)
    db1 = zeros(Cr, dev) if has_bias else None
    db2 = zeros(C, dev) if has_bias else None
    call("mfnet_se_gate_bwd", dg.data_ptr(), pooled_sum.data_ptr(), P, W1.data_ptr(), W2.data_ptr(), 1 if relu6 else 0, z.data_ptr(),
         g.data_ptr(), dpool.data_ptr(), dW1.data_ptr(), _p(db1), dW2.data_ptr(), _p(db2), B, C, Cr, _stream(W1))
    return dpool, dW1, db1, dW2, db2


def se_scale_sum(rx, ry, gx, gy, B, P, out_dtype=torch.bfloat16):
    _chk(rx, ry, gx, gy)
    C = rx.shape[-1]
    out = torch.empty((B * P, C), dtype=_odt(out_dtype), device=rx.device)
    call("mfnet_se_scale_sum", rx.data_ptr(), _dt(rx), ry.data_ptr(), _dt(ry), gx.data_ptr(), gy.data_ptr(), out.data_ptr(), _dt(out), B, P, C, _stream(rx))
    return out


def se_bwd_apply(dres, gx, gy, dpx, dpy, B, P, dtype_x=None, dtype_y=None, out_x=None, out_y=None):
    _chk(dres)
    C = dres.shape[-1]
    assert dres.is_contiguous()
    drx = out_x if out_x is not None else torch.empty((B * P, C), dtype=_odt(dtype_x), device=dres.device)
    dry = out_y if out_y is not None else torch.empty((B * P, C), dtype=_odt(dtype_y), device=dres.device)
    assert drx.is_contiguous() and dry.is_contiguous()
    call("mfnet_se_bwd_apply", dres.data_ptr(), _dt(dres), gx.data_ptr(), gy.data_ptr(), dpx.data_ptr(), dpy.data_ptr(), drx.data_ptr(),
         _dt(drx), dry.data_ptr(), _dt(dry), B, P, C, _stream(dres))
    return drx, dry


def frh_gate_fwd(xf, dwo, gate, B, P):
    _chk(xf, dwo, gate)
    pc = torch.empty_like(xf)
    call("mfnet_frh_gate_fwd", xf.data_ptr(), dwo.data_ptr(), gate.data_ptr(), pc.data_ptr(), B, P, xf.shape[-1], _stream(xf))
    return pc


def frh_gate_bwd(dpc, xf, dwo, gate, dpool, B, P):
    _chk(dpc, xf, dwo, gate, dpool)
    assert dpc.is_contiguous() and dpc.dtype == torch.float32
    dxf = torch.empty_like(xf)
    ddwo = torch.empty_like(xf)
    call("mfnet_frh_gate_bwd", dpc.data_ptr(), xf.data_ptr(), dwo.data_ptr(), gate.data_ptr(), _p(dpool), dxf.data_ptr(), ddwo.data_ptr(),
         B, P, xf.shape[-1], _stream(xf))
    return dxf, ddwo


def add3(a, b, c=None, out_dtype=torch.float32, out=None):
    _chk(a, b, c)
    assert a.is_contiguous() and b.is_contiguous() and (c is None or c.is_contiguous())
    if out is None:
        out = torch.empty(a.shape, dtype=out_dtype, device=a.device)
    call("mfnet_add3", a.data_ptr(), _dt(a), b.data_ptr(), _dt(b), _p(c), _dt(c) if c is not None else 0, out.data_ptr(), _dt(out), a.numel(), _stream(a))
    return out


# ------------------------------------------------------------------------------------------ raster inference
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize_rgb_u8(img_u8, scale=1.0 / 255.0, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """[3,H,W] uint8 -> fp32 (x*scale - mean)/std  (inference.py:464-484)."""
    import ctypes
    _chk(img_u8)
    assert img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and img_u8.shape[0] == 3
    out = torch.empty(img_u8.shape, dtype=torch.float32, device=img_u8.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    call("mfnet_normalize_rgb_u8", img_u8.data_ptr(), out.data_ptr(), img_u8.shape[1], img_u8.shape[2], float(scale),
         ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), _stream(img_u8))
    return out


def extract_tiles(rgb, dsm, coords4, h, w, check_extents=False):
    """rgb [C,H,W] fp32, dsm [H,W] fp32, coords4 [n,4] int32 (row, col, valid_h, valid_w) -> ([n,C,h,w], [n,h,w])."""
    _chk(rgb, dsm, coords4)
    assert rgb.is_contiguous() and dsm.is_contiguous() and rgb.dtype == dsm.dtype == torch.float32
    assert coords4.dtype == torch.int32 and coords4.is_contiguous() and coords4.shape[1] == 4
    C, H, W = rgb.shape
    n = coords4.shape[0]
    if check_extents and n:
        # the kernel trusts the extents (coords live on the device): row + valid_h <= H, col + valid_w <= W, 1 <= valid <= tile
        c = coords4.cpu() if coords4.is_cuda else coords4
        ok = ((c[:, 0] >= 0) & (c[:, 1] >= 0) & (c[:, 2] >= 1) & (c[:, 3] >= 1) & (c[:, 2] <= h) & (c[:, 3] <= w)
              & (c[:, 0] + c[:, 2] <= H) & (c[:, 1] + c[:, 3] <= W))
        if not bool(ok.all()):
            raise MfnetError("extract_tiles: a window leaves the raster (row + valid_h <= H, col + valid_w <= W, 1 <= valid <= tile)")
    ro = torch.empty((n, C, h, w), dtype=torch.float32, device=rgb.device)
    do = torch.empty((n, h, w), dtype=torch.float32, device=rgb.device)
    call("mfnet_extract_tiles", rgb.data_ptr(), dsm.data_ptr(), coords4.data_ptr(), ro.data_ptr(), do.data_ptr(), n, C, H, W, h, w, _stream(rgb))
    return ro, do


def tile_minmax_norm(tiles, keep_flat=False):
    _chk(tiles)
    assert tiles.is_contiguous() and tiles.dtype == torch.float32
    call("mfnet_tile_minmax_norm", tiles.data_ptr(), tiles.shape[0], tiles[0].numel(), 1 if keep_flat else 0, _stream(tiles))
    return tiles


def prepare_tiles(rgb_u8, dsm, labels, codes, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """rgb uint8 [n,3,S,S], dsm fp32 [n,S,S] | None, labels int64 [n,S,S] | None, codes int32 [n] (bit 0 hflip,
    bit 1 vflip, bits 2-3 rot90 k) -> (rgb fp32 normalised, dsm, labels), all transformed alike."""
    import ctypes
    _chk(rgb_u8, dsm, labels, codes)
    n, _, S, S2 = rgb_u8.shape
    assert S == S2 and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and codes.dtype == torch.int32 and codes.numel() == n
    assert dsm is None or (dsm.dtype == torch.float32 and dsm.is_contiguous() and tuple(dsm.shape) == (n, S, S))
    assert labels is None or (labels.dtype == torch.int64 and labels.is_contiguous() and tuple(labels.shape) == (n, S, S))
    ro = torch.empty((n, 3, S, S), dtype=torch.float32, device=rgb_u8.device)
    do = torch.empty_like(dsm) if dsm is not None else None
    lo = torch.empty_like(labels) if labels is not None else None
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    call("mfnet_prepare_tiles", rgb_u8.data_ptr(), _p(dsm), _p(labels), codes.data_ptr(), ro.data_ptr(), _p(do), _p(lo), n, S,
         ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), _stream(rgb_u8))
    return ro, do, lo


def jitter_tiles(rgb_u8, params):
    """Brightness / contrast jitter in place on uint8 patches [n,3,S,S] (multiscale_dataset.py:250-265); params fp32 [n,2] on the
    device = (brightness, contrast) per tile, brightness <= 0: tile untouched."""
    _chk(rgb_u8, params)
    assert rgb_u8.is_contiguous() and rgb_u8.dtype == torch.uint8 and rgb_u8.dim() == 4 and rgb_u8.shape[1] == 3 and rgb_u8.shape[2] == rgb_u8.shape[3]
    assert params.is_contiguous() and params.dtype == torch.float32 and tuple(params.shape) == (rgb_u8.shape[0], 2)
    tcall("mfnet_jitter_tiles", rgb_u8.numel() * 3, 0, rgb_u8.data_ptr(), params.data_ptr(), rgb_u8.shape[0], rgb_u8.shape[2], _stream(rgb_u8))
    return rgb_u8


def tile_prob_accumulate(psum, cnt, logits, coords4, cls):
    """psum / cnt: int64 [H, W] (psum in fixed point, probability * 2^40): integer atomics, order-independent."""
    _chk(psum, cnt, logits, coords4)
    assert psum.is_contiguous() and cnt.is_contiguous() and logits.is_contiguous() and logits.dtype == torch.float32
    assert psum.dtype == torch.int64 and cnt.dtype == torch.int64
    H, W = psum.shape
    n, C, h, w = logits.shape
    call("mfnet_tile_prob_accumulate", psum.data_ptr(), cnt.data_ptr(), logits.data_ptr(), coords4.data_ptr(), n, C, H, W, h, w, cls, _stream(psum))


def blend_finish(psum, cnt, threshold):
    _chk(psum, cnt)
    assert psum.dtype == torch.int64 and cnt.dtype == torch.int64 and psum.is_contiguous() and cnt.is_contiguous()
    prob = torch.empty(psum.shape, dtype=torch.float32, device=psum.device)
    mask = torch.empty(psum.shape, dtype=torch.uint8, device=psum.device)
    call("mfnet_blend_finish", psum.data_ptr(), cnt.data_ptr(), prob.data_ptr(), mask.data_ptr(), float(threshold), psum.numel(), _stream(psum))
    return prob, mask


def percentiles_2_98(values):
    """np.percentile(values, (2, 98)) ('linear' method) of a 1-D fp32 device tensor: one device sort, the two
    neighbours of each virtual index are interpolated on the host the way numpy's _lerp does (inference.py:474)."""
    import numpy as np
    v = torch.sort(values.reshape(-1).float()).values
    n = v.numel()
    out = []
    for q in (2.0, 98.0):
        vi = (q / 100.0) * (n - 1)
        lo = min(max(int(np.floor(vi)), 0), n - 1)
        hi = min(lo + 1, n - 1)
        a, b = (np.float32(x) for x in v[[lo, hi]].tolist())
        t = vi - lo
        d = np.float64(np.float32(b - a))                  # numpy's _lerp takes b - a in the array's dtype
        r = np.float64(a) + d * t if t < 0.5 else np.float64(b) - d * (1 - t)
        out.append(float(r))
    return out[0], out[1]


def normalize_raster(image, is_rgb=True, nodata_value=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """BuildingInference._normalize_image (inference.py:444-496) on the device. image: [C,H,W] or [H,W], uint8 / uint16 /
    floating -> (fp32 image of the same shape, bool nodata mask [H,W])."""
    import ctypes
    _chk(image)
    img = image.contiguous()
    chw = img if img.dim() == 3 else img.unsqueeze(0)
    C, H, W = chw.shape
    kind = {torch.uint8: 0, torch.uint16: 1}.get(chw.dtype, 2)
    if kind == 2 and chw.dtype != torch.float32:
        chw = chw.float()
    mask = torch.empty((H, W), dtype=torch.uint8, device=img.device)
    out = torch.empty((C, H, W), dtype=torch.float32, device=img.device)
    has_nd = nodata_value is not None
    nd = float(nodata_value) if has_nd else 0.0
    if has_nd and kind == 2:
        import numpy as np
        nd = float(np.float32(nd))                                  # numpy compares an fp32 array with the scalar as fp32
    clip, sub, div = 0, 0.0, 1.0
    if is_rgb:
        if kind == 0:
            div = 255.0
        elif kind == 1:
            div = 65535.0
        else:                                                          # :471-477 percentile stretch over the valid pixels
            clip = 1
            if has_nd:
                valid = ~(chw == nd).any(dim=0)
                vals = chw[:, valid]
            else:
                vals = chw
            if vals.numel() > 0:
                import numpy as np
                p2, p98 = percentiles_2_98(vals)
                # numpy 1.24 (the reference's pin) casts the float64 scalars to the array's fp32 (value-based casting)
                sub, div = float(np.float32(p2)), float(np.float32(p98 - p2))
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    call("mfnet_normalize_raster", chw.data_ptr(), kind, C, H, W, 1 if is_rgb else 0, clip, float(sub), float(div),
         ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), 1 if has_nd else 0, nd, out.data_ptr(), mask.data_ptr(),
         _stream(img))
    return (out if img.dim() == 3 else out[0]), mask.bool()


def tile_prob_mask(logits, coords4, cls, threshold, want_prob=False):
    """logits [n,C,h,w] fp32, coords4 [n,4] int32 -> (prob [n,h,w] | None, mask uint8 [n,h,w]); 0 outside each tile's extent."""
    _chk(logits, coords4)
    assert logits.is_contiguous() and logits.dtype == torch.float32 and coords4.dtype == torch.int32 and coords4.is_contiguous()
    n, C, h, w = logits.shape
    assert coords4.shape[0] == n
    prob = torch.empty((n, h, w), dtype=torch.float32, device=logits.device) if want_prob else None
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=logits.device)
    call("mfnet_tile_prob_mask", logits.data_ptr(), coords4.data_ptr(), _p(prob), mask.data_ptr(), n, C, h, w, int(cls), float(threshold), _stream(logits))
    return prob, mask


def confusion(pred_u8, gt_i64, n_classes):
    _chk(pred_u8, gt_i64)
    assert pred_u8.dtype == torch.uint8 and gt_i64.dtype == torch.int64 and pred_u8.is_contiguous() and gt_i64.is_contiguous()
    assert pred_u8.numel() == gt_i64.numel()
    cm = torch.zeros((n_classes, n_classes), dtype=torch.int64, device=pred_u8.device)
    call("mfnet_confusion", pred_u8.data_ptr(), gt_i64.data_ptr(), cm.data_ptr(), n_classes, pred_u8.numel(), _stream(pred_u8))
    return cm


# ------------------------------------------------------------------------------------------ batched strided copies
def gather(pairs, accumulate=False):
    """One launch (per 48 pairs) for many small strided copies: pairs = [(dst_view, src_view)] with equal
    logical shapes of at most 4 dims (expanded / permuted / sliced views welcome), src fp32, dst fp32 (assign,
    or += with accumulate) or bf16 (assign); all dst of one call share a dtype."""
    import ctypes
    if not pairs:
        return
    import struct
    ddt = pairs[0][0].dtype
    desc = (ctypes.c_int64 * (17 * len(pairs)))()
    k = 0
    nbytes = 0
    for dst, src in pairs:
        red_n, red_stride, scale = 1, 0, 1.0
        if isinstance(src, SplitSum):
            red_n, red_stride, scale, src = src.part.shape[0], src.part.stride(0), src.scale, src.part[0]
        _chk(dst, src)
        assert src.dtype == torch.float32 and dst.dtype == ddt and tuple(dst.shape) == tuple(src.shape), (dst.shape, src.shape)
        nd = dst.dim()
        assert nd <= 4
        pad = 4 - nd
        desc[k] = src.data_ptr()
        desc[k + 1] = dst.data_ptr()
        for i in range(4):
            j = i - pad
            desc[k + 2 + i] = dst.shape[j] if j >= 0 else 1
            desc[k + 6 + i] = src.stride(j) if j >= 0 else 0
            desc[k + 10 + i] = dst.stride(j) if j >= 0 else 0
        desc[k + 14], desc[k + 15] = red_n, red_stride
        desc[k + 16] = struct.unpack("<I", struct.pack("<f", scale))[0]
        k += 17
        nbytes += dst.numel() * (4 * red_n + dst.element_size() * (2 if accumulate else 1))
    tcall("mfnet_gather_segments", nbytes, 0, ctypes.cast(desc, ctypes.c_void_p), len(pairs), _DT[ddt], 1 if accumulate else 0, _stream(pairs[0][0]))
