#!/usr/bin/env python3
"""Micro-benchmark of the LoRA forward fusions at the bench shapes: python tools/ln_lora_bench.py"""
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


M = 8192
x = torch.randn(M, 768, device="cuda")
g, b = torch.randn(768, device="cuda"), torch.randn(768, device="cuda")
A8 = torch.randn(2, 8, 768, device="cuda")
A4 = torch.randn(2, 4, 768, device="cuda")
wx, wy = torch.tensor([0.7], device="cuda"), torch.tensor([0.6], device="cuda")
print(f"ln_fwd_lora  R=8        : {timeit(lambda: ops.layernorm_fwd_lora(x, g, b, 1e-6, A8)):6.1f} us")
print(f"ln_fwd_lora  R=4 + mix  : {timeit(lambda: ops.layernorm_fwd_lora(x, g, b, 1e-6, A4, wx, wy)):6.1f} us")
print(f"layernorm_fwd (plain)   : {timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6)):6.1f} us")
h = torch.randn(M, 3072, device="cuda").bfloat16()
A2 = torch.randn(2, 4, 3072, device="cuda")
print(f"lora_down_mix C=3072    : {timeit(lambda: ops.lora_down_mix(h, A2, wx, wy)):6.1f} us")
