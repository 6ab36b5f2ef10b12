"""Per-step operands derived from TRAINABLE weights, and direct delivery of parameter gradients.

Both exist to keep the train step free of the ~700 tiny framework launches (casts, permuted copies, stacks,
`cat`s, and autograd's per-parameter `grad += g`) that the reference's step performs implicitly
(trainer.py:146-197 around unetformer_mmsam.py / loralib modules). Neither changes any arithmetic.

ParamBank - the engine owns one. While it is active (inside TrainEngine.forward_backward), `derived(key, ...)`
returns a persistent tensor that `ParamBank.refresh()` re-fills from the current parameter values in a few
`mfnet_gather_segments` launches at the start of every step (inside the captured HIP graph in graph mode).
The first time a key is seen it is built on the spot and registered. Without an active bank and with autograd on
(plain module calls, autograd tests) every call builds a fresh tensor; under no_grad (inference, evaluation between
steps) the tensor is kept and validated against its sources' version counters and WEIGHTS_EPOCH on every call. What that
check sees: every in-place torch op on a parameter (load_state_dict, optimizer steps, `p.mul_()`), and whoever calls
`weights_changed()`. What it does NOT see: writes through `p.data` (`p.data.mul_()`, `p.data.copy_()` - `.data` is a tensor
with a version counter of its own) - such code must call `weights_changed()` itself, as TrainEngine's SGD kernel and broadcast do -
and a parameter re-pointed to new storage (`p.data = new`, `model.to()`): the stored source views keep the old storage alive, so
the comparison passes; `inference.GraphedForward` fingerprints the parameter addresses before every replay for that case and
drops its graphs and this store when they moved.

deliver - a parameter the engine adopted carries `_mfnet_sink = (grad_view, notify)`: its gradient, in
whatever strided layout the kernels produced it, is added into the flat gradient buffer by the same gather
kernel (one launch per module backward) and autograd receives None for it. Other parameters get an ordinary
gradient tensor back.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import ops

Pair = Tuple[torch.Tensor, torch.Tensor]


class ParamBank:
    active: Optional["ParamBank"] = None

    def __init__(self):
        self.entries = {}
        # defer: gradient deliveries of a step are collected and added into the flat gradient buffer by a few gather
        # launches when the step's backward pass is over (one launch per 48 tensors instead of one per module backward).
        # Off where a collective is launched from inside the backward pass and must find its bucket's gradients in place.
        self.defer = False
        self._pending, self._notes = [], []

    def flush(self):
        ops.tn_flush()              # the weight-gradient products collected during backward write the partial tiles gathered below
        if self._pending:
            ops.gather(self._pending, accumulate=True)
            for notify, p in self._notes:
                notify(p)
            self._pending, self._notes = [], []

    def refresh(self):
        by_dtype = {}
        for _, pairs in self.entries.values():
            for d, s in pairs:
                by_dtype.setdefault(d.dtype, []).append((d, s))
        for pairs in by_dtype.values():
            ops.gather(pairs)

    def __enter__(self):
        ParamBank.active = self
        # deferred delivery: nothing reads a weight gradient before flush(), so the products that compute them are collected too
        ops.TN_QUEUE = [] if (self.defer and os.environ.get("MFNET_TN_BATCH", "1") != "0") else None      # (=0: the A/B switch)
        self.refresh()
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.flush()
        else:
            self._pending, self._notes = [], []
        ops.TN_QUEUE = None
        ParamBank.active = None
        ops._twin[0] = None
        return False


def derived(key, shape, dtype, fill: Callable[[torch.Tensor], List[Pair]], device=None) -> torch.Tensor:
    """Tensor of `shape` whose content is described by `fill(t) -> [(view of t, fp32 source view)]`; parts of
    t no pair covers stay zero. `key` must identify the sources (use their data_ptr()s) and the recipe."""
    if ops.FP32 and dtype == torch.bfloat16:         # fp32 training: the GEMM image of a weight is the fp32 weight in the same layout
        dtype, key = torch.float32, ("fp32",) + tuple(key)
    bank = ParamBank.active
    if bank is not None:
        hit = bank.entries.get(key)
        if hit is not None:
            return hit[0]
    elif EvalBank.active is not None and not torch.is_grad_enabled():
        # Inference inside a predictor's EvalBank: the operand is kept and re-used while its sources are unchanged - their
        # version counters (bumped by every in-place torch op, load_state_dict included) and WEIGHTS_EPOCH (bumped by whoever
        # rewrites parameters behind torch's back: TrainEngine's SGD kernel, its broadcast, a training-mode BatchNorm) are
        # compared on every call. A sliding-window forward re-derived ~240 operands per batch before (15 % of its kernel time
        # in gather / fill launches). Keys carry data_ptr()s, so the store belongs to ONE predictor, whose forward keeps its
        # model alive (a process-wide store could hand a freed model's operand to a new model at the same address). Under
        # graph capture a stored operand is used as it is (GraphedForward validates the store before every replay and
        # refreshes changed operands in place); a key first seen under capture is built inside the graph and not stored.
        eb = EvalBank.active
        capturing = torch.cuda.is_current_stream_capturing()
        hit = eb.entries.get(key)
        if hit is not None and tuple(hit[0].shape) == tuple(shape) and hit[0].dtype == dtype:
            if not capturing:
                eb.check(key, hit)
            return hit[0]
        if not capturing:
            t = torch.zeros(shape, dtype=dtype, device=device)
            pairs = fill(t)
            ops.gather(pairs)
            eb.entries[key] = [t, pairs, _signature(pairs)]
            return t
    t = torch.zeros(shape, dtype=dtype, device=device)
    with torch.no_grad():
        pairs = fill(t)
        ops.gather(pairs)
    if bank is not None:
        bank.entries[key] = (t, pairs)
    return t


WEIGHTS_EPOCH = ops.WEIGHTS_EPOCH          # one counter for every cache of weight-derived tensors (ops.bn_forward's included)


class EvalBank:
    """Store of weight-derived operands for inference: a predictor owns one and runs its forwards inside it."""
    active: Optional["EvalBank"] = None

    def __init__(self):
        self.entries, self.bn = {}, {}

    def __enter__(self):
        self._prev = (EvalBank.active, ops._BN_EVAL)
        EvalBank.active, ops._BN_EVAL = self, self.bn
        return self

    def __exit__(self, *exc):
        EvalBank.active, ops._BN_EVAL = self._prev
        return False

    def check(self, key, ent):
        """re-fill the operand in place if one of its sources changed since it was written"""
        sig = _signature(ent[1])
        if sig != ent[2]:
            ent[0].zero_()
            ops.gather(ent[1])
            ent[2] = sig

    def validate(self):
        """every stored operand against its sources (before a graph replay that reads them): a few hundred integer
        compares; changed ones are re-filled in place, so a captured forward sees the new values at the same addresses"""
        for key, ent in self.entries.items():
            self.check(key, ent)
        for key, ent in self.bn.items():
            ops.bn_eval_check(ent)


def weights_changed():
    """to be called by code that rewrites parameters without a torch in-place op (custom kernels on a flat buffer)"""
    WEIGHTS_EPOCH[0] += 1


def _signature(pairs):
    return (WEIGHTS_EPOCH[0],) + tuple((s.data_ptr(), s._version) for _, s in pairs)


def weight_bf16(W2d: torch.Tensor, kind="w") -> torch.Tensor:
    """bf16 GEMM image of a strided fp32 view of a trainable weight (<=4-D; e.g. w.view(O, I).t())."""
    W2d = W2d.detach()
    key = (kind, W2d.data_ptr(), tuple(W2d.shape), tuple(W2d.stride()))
    return derived(key, tuple(W2d.shape), torch.bfloat16, lambda t: [(t, W2d)], device=W2d.device)


def _real(g):
    return g.materialize() if isinstance(g, ops.SplitSum) else g


def deliver(params: Sequence[torch.Tensor], grads: Sequence) -> tuple:
    """-> the gradient tuple to return to autograd for `params`. grads[i] is None, a tensor view shaped like
    params[i] (any strides), or a list of (row_slice, tensor) pieces that tile params[i] along dim 0."""
    out, pairs, notes = [], [], []
    if ops.TN_QUEUE:
        bank = ParamBank.active
        if bank is None or not bank.defer or any(g is not None and getattr(p, "_mfnet_sink", None) is None for p, g in zip(params, grads)):
            ops.tn_flush()          # a gradient leaves as a tensor, or is added right now: the collected products must have run
    for p, g in zip(params, grads):
        sink = getattr(p, "_mfnet_sink", None)
        if g is None:
            out.append(None)
        elif sink is not None:
            view, notify = sink
            if isinstance(g, list):
                for sl, piece in g:
                    pairs.append((view[sl], piece))
            else:
                pairs.append((view, g.view(p.shape) if tuple(g.shape) != tuple(p.shape) else g))
            if notify is not None:
                notes.append((notify, p))
            out.append(None)
        elif isinstance(g, list):
            out.append(torch.cat([_real(piece) for _, piece in g], 0).view(p.shape))
        else:
            out.append(_real(g).reshape(p.shape).contiguous())
    bank = ParamBank.active
    if pairs and bank is not None and bank.defer:
        bank._pending += pairs
        bank._notes += notes
    elif pairs:
        ops.gather(pairs, accumulate=True)
        for notify, p in notes:
            notify(p)
    return tuple(out)
