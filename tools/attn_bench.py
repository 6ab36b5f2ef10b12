#!/usr/bin/env python3
"""Micro-benchmark of the SAM attention kernels at the bench configuration (32 image-modalities of 16x16
tokens, 12 heads of 64): global and 14x14-windowed, forward and backward (random operands).
    python tools/attn_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops  # noqa: E402
from ssrs_amd.vit import build_rel_table  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--G", type=int, default=32)
    ap.add_argument("--S", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    G, S, heads, hd = a.G, a.S, 12, 64
    D = heads * hd
    qkv = (torch.randn(G * S * S, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.randn(3 * D, device=dev) * 0.1
    dout = (torch.randn(G * S * S, D, device=dev) * 0.1).to(torch.bfloat16)
    for ws in (0, 14):
        k = ws if ws else S
        tbl, tblT, Lh, Lw = build_rel_table(torch.randn(2 * k - 1, hd, device=dev) * 0.1, torch.randn(2 * k - 1, hd, device=dev) * 0.1, k, k)
        out, lse = ops.attn_fwd(qkv, bias, tbl, G, S, S, heads, ws, Lh, Lw, hd ** -0.5)
        tf = timeit(lambda: ops.attn_fwd(qkv, bias, tbl, G, S, S, heads, ws, Lh, Lw, hd ** -0.5), a.reps)
        tb = timeit(lambda: ops.attn_bwd(qkv, bias, tbl, tblT, out, dout, lse, G, S, S, heads, ws, Lh, Lw, hd ** -0.5), a.reps)
        print(f"ws={ws:2d}: fwd {tf:7.1f} us   bwd (dq + dkv) {tb:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
