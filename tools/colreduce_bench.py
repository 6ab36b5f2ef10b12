#!/usr/bin/env python3
"""Micro-benchmark of the column reductions (BatchNorm statistics / SE pools) at decoder shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for G, P, C, dt in [(1, 65536, 64, torch.float32), (1, 16384, 256, torch.float32), (1, 65536, 256, torch.bfloat16), (1, 4096, 64, torch.float32),
                    (1, 1024, 64, torch.float32), (16, 4096, 64, torch.float32), (16, 4096, 256, torch.bfloat16), (16, 256, 256, torch.float32)]:
    x = torch.randn(G * P, C, device="cuda").to(dt)
    t = timeit(lambda: ops.colsum(x, G, P, want_sq=True))
    mb = x.numel() * x.element_size() / 1e6
    print(f"colsum G={G:3d} P={P:6d} C={C:4d} {str(dt)[6:]:9s} {t:7.1f} us  {mb / t * 1e-3:6.2f} TB/s (incl. the zero fill)", flush=True)
