#!/usr/bin/env python3
"""Developer check of the 256-row GEMM kernel: every tile shape x epilogue against the 128x128 kernel (same arithmetic order per
output element up to the k-tile split -> equal to rounding) and fp64, then per-shape timings old vs new.
    python tools/gemm_big_check.py [--bench-only]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops  # noqa: E402


def mk(M, N, K, kind, dev):
    x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * K ** -0.5).to(torch.bfloat16)
    lora = dict(U=torch.rand(M, 8, device=dev), V=torch.rand(2, N, 8, device=dev), rows_per_group=M // 2, n_groups=2, lr_scale=0.25)
    bias = torch.rand(N, device=dev)
    kw = dict(plain={}, bias=dict(bias=bias), lin1=dict(bias=bias, act=1, want_pre=True, **lora),
              lin2d=dict(act=2, aux=torch.randn(M, N, device=dev).to(torch.bfloat16), **lora), qkv=dict(bias=bias, **lora),
              lin2=dict(bias=bias, residual=torch.rand(M, N, device=dev), out_dtype=torch.float32, **lora),
              proj=dict(bias=bias, residual=torch.rand(M, N, device=dev), out_dtype=torch.float32),
              relu6=dict(bias=bias, act=3), relu6g=dict(act=4, aux=torch.randn(M, N, device=dev).to(torch.bfloat16) * 4),
              res5=dict(act=5, residual=torch.rand(M, N, device=dev), out_dtype=torch.float32),
              r16=dict(U=torch.rand(M, 16, device=dev), V=torch.rand(N, 16, device=dev), lr_scale=0.5, out_dtype=torch.float32),
              r4=dict(U=torch.rand(M, 4, device=dev), V=torch.rand(2, N, 4, device=dev), rows_per_group=M // 2, n_groups=2, lr_scale=0.25))[kind]
    return x, w, kw


def first(o):
    return o if not isinstance(o, tuple) else o[0]


def check():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    bad = 0
    for nf in (3, 4, 6, 8, 9):
        N = 32 * nf * 2
        for K in (64, 192, 768):
            for kind in ("plain", "bias", "lin1", "lin2d", "qkv", "lin2", "proj", "relu6", "relu6g", "res5", "r16", "r4"):
                M = 512
                x, w, kw = mk(M, N, K, kind, dev)
                ref = ops.gemm(x, w, tile=-1, **kw)
                got = ops.gemm(x, w, tile=nf, **kw)
                outs = zip(ref if isinstance(ref, tuple) else (ref,), got if isinstance(got, tuple) else (got,))
                for r, g in outs:
                    err = ((r.float() - g.float()).abs().max() / r.float().abs().max().clamp_min(1e-20)).item()
                    tol = 1e-2 if r.dtype == torch.bfloat16 else 2e-5
                    if not err < tol:
                        bad += 1
                        print(f"MISMATCH nf={nf} K={K} {kind}: rel err {err:.3e}")
        # exact check with A = I, asymmetric W
        K = 32 * nf * 2
        a = torch.zeros((K + 255) // 256 * 256, K, device=dev)
        a[:K, :K] = torch.eye(K, device=dev)
        wi = (torch.arange(N * K, device=dev).reshape(N, K) % 251).float().to(torch.bfloat16)
        if K % 64 == 0:
            out = ops.gemm(a.to(torch.bfloat16), wi, out_dtype=torch.float32, tile=nf)
            if not torch.equal(out[:K], wi.float().t().contiguous()):
                bad += 1
                print(f"IDENTITY MISMATCH nf={nf}")
    print("check:", "OK" if not bad else f"{bad} mismatches")
    return bad


def bench(reps=30):
    dev = torch.device("cuda")
    shapes = [(8192, 3072, 768, "lin1", (-1, 6, 4, 8)), (8192, 3072, 768, "lin2d", (-1, 6, 4, 8)), (8192, 3072, 768, "plain", (-1, 6, 4, 8)),
              (8192, 768, 3072, "lin2", (-1, 3, 4, 6)), (8192, 768, 3072, "qkv", (-1, 3, 4, 6)), (8192, 768, 3072, "plain", (-1, 3, 4)),
              (8192, 768, 768, "proj", (-1, 3, 4, 6)), (8192, 768, 768, "plain", (-1, 3, 4)),
              (8192, 2304, 768, "qkv", (-1, 9, 6, 4, 3)), (8192, 2304, 768, "plain", (-1, 9, 6, 4)), (8192, 768, 2304, "qkv", (-1, 3, 4)),
              (4096, 4096, 4096, "plain", (-1, 8, 4)), (16384, 1024, 256, "plain", (-1, 4, 8)), (16384, 256, 1024, "plain", (-1, 4, 8))]
    for M, N, K, kind, tiles in shapes:
        x, w, kw = mk(M, N, K, kind, dev)
        res = []
        for tile in tiles:
            if tile > 0 and N % (32 * tile):
                continue
            for _ in range(3):
                ops.gemm(x, w, tile=tile, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm(x, w, tile=tile, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            res.append(f"tile {tile:2d}: {us:6.1f} us {2 * M * N * K / us / 1e6:6.0f} TF")
        print(f"M,N,K=({M},{N},{K}) {kind:6s} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    if "--bench-only" not in sys.argv:
        if check():
            sys.exit(1)
    bench()
