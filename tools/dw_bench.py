#!/usr/bin/env python3
"""Micro-benchmark of the decoder's depthwise convs at the bench shapes: python tools/dw_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for B, S, k, ext in [(16, 8, 8, True), (16, 16, 8, True), (16, 32, 8, True), (16, 64, 3, False)]:
    C = 64
    x = torch.randn(B, S, S, C, device="cuda")
    w = torch.randn(C, k * k, device="cuda")
    dy = torch.randn(B, S, S, C, device="cuda")
    tf = timeit(lambda: ops.dwconv_fwd(x, w, None, B, S, S, k, ext))
    dx = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    td = timeit(lambda: ops.call("mfnet_dwconv_bwd_data", dy.data_ptr(), 0, w.data_ptr(), dx.data_ptr(), B, S, S, C, k, 1 if ext else 0, st))
    dw = torch.zeros(32, C, k * k, device="cuda")
    tw = timeit(lambda: ops.call("mfnet_dwconv_bwd_weight", dy.data_ptr(), 0, x.data_ptr(), dw.data_ptr(), None, B, S, S, C, k, 1 if ext else 0, 32, st))
    print(f"B={B} {S}x{S} k={k}: fwd {tf:6.1f} us  bwd_data {td:6.1f} us  bwd_weight {tw:6.1f} us", flush=True)
