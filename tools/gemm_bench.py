#!/usr/bin/env python3
"""Micro-benchmark of mfnet_gemm_bf16 on the encoder's GEMM shapes (random operands).
    python tools/gemm_bench.py [--reps 30]
Prints per shape: microseconds per launch and algorithmic TFLOP/s (HIP events on the launch stream)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops  # noqa: E402

SHAPES = [(8192, 3072, 768, "lin1 fwd: bias+lora+gelu+pre"), (8192, 3072, 768, "lin2 dgrad: lora+gelu'"),
          (8192, 2304, 768, "qkv fwd: bias+lora"), (8192, 768, 3072, "lin2 fwd: bias+lora+res"), (8192, 3072, 768, "gelu only"),
          (8192, 2304, 768, "bias only"), (8192, 2304, 768, "lora only g1"), (8192, 2304, 768, "lora only g2"), (8192, 3072, 768, "plain"), (8192, 768, 3072, "plain"),
          (8192, 768, 3072, "residual f32"), (8192, 2304, 768, "plain"), (8192, 768, 2304, "plain"),
          (8192, 768, 768, "plain"), (4096, 4096, 4096, "plain"), (65536, 64, 576, "plain"), (16384, 256, 1024, "plain")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rotate", type=int, default=0, help="keep the last N results alive and cycle N copies of the activation operand: every "
                    "launch reads and writes fresh addresses (N x footprint > the 256 MB MALL), like the GEMMs inside a train step")
    ap.add_argument("--only", type=int, default=0, help="only the first N shapes")
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    for M, N, K, kind in (SHAPES[:a.only] if a.only else SHAPES):
        x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        kw = {}
        lora = dict(U=torch.rand(M, 8, device=dev), V=torch.rand(2, N, 8, device=dev), rows_per_group=M // 2, n_groups=2, lr_scale=0.25)
        if kind.startswith("lin1"):
            kw = dict(bias=torch.rand(N, device=dev), act=1, want_pre=True, **lora)
        elif kind.startswith("bias only"):
            kw = dict(bias=torch.rand(N, device=dev))
        elif kind.startswith("lora only g1"):
            kw = dict(U=lora["U"], V=lora["V"][0].contiguous(), lr_scale=0.25)
        elif kind.startswith("lora only g2"):
            kw = dict(**lora)
        elif kind.startswith("gelu only"):
            kw = dict(act=1)
        elif kind.startswith("lin2 dgrad"):
            kw = dict(act=2, aux=torch.randn(M, N, device=dev).to(torch.bfloat16), **lora)
        elif kind.startswith("qkv fwd"):
            kw = dict(bias=torch.rand(N, device=dev), **lora)
        elif kind.startswith("lin2 fwd"):
            kw = dict(bias=torch.rand(N, device=dev), residual=torch.rand(M, N, device=dev), out_dtype=torch.float32, **lora)
        elif kind.startswith("residual"):
            kw = dict(residual=torch.rand(M, N, device=dev), out_dtype=torch.float32)
        out = ops.gemm(x, w, **kw)
        o = out[0] if isinstance(out, tuple) else out
        if kind == "plain":
            ref = x[:256].float() @ w.float().t()
            err = (o[:256].float() - ref).abs().max().item() / ref.abs().max().item()
        else:
            err = float("nan")
        for _ in range(3):
            ops.gemm(x, w, **kw)
        xs = [x.clone() for _ in range(a.rotate)] or [x]
        kws = [dict(kw) for _ in xs]
        for k2 in kws[1:]:
            for name in ("aux", "residual"):
                if name in k2:
                    k2[name] = k2[name].clone()
        ring = []
        for i in range(2 * len(xs)):
            ring.append(ops.gemm(xs[i % len(xs)], w, **kws[i % len(xs)]))
            if len(ring) > a.rotate:
                ring.pop(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.reps):
            ring.append(ops.gemm(xs[i % len(xs)], w, **kws[i % len(xs)]))
            if len(ring) > a.rotate:
                ring.pop(0)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / a.reps
        print(f"M,N,K=({M},{N},{K}) {kind:26s} {us:8.1f} us  {2 * M * N * K / us / 1e6:8.1f} TF/s  relerr {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
