#!/usr/bin/env python3
"""Micro-benchmark: the fused LoRA backward side pass (mfnet_lora_side) against the two kernels it replaces (skinny_nt + skinny_tn)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops  # noqa: E402


def t(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


dev = torch.device("cuda")
M = 8192
for C, G in ((768, 1), (3072, 1), (2304, 2)):
    y = torch.randn(M, C, device=dev).to(torch.bfloat16)
    U = torch.randn(M, 8, device=dev)
    Vt = torch.randn((G, 8, C) if G > 1 else (8, C), device=dev)
    da = torch.randn(M, 4 if G == 1 else 8, device=dev)
    mb = M * C * 2 / 1e6
    a = t(lambda: ops.skinny_nt(y, Vt, rows_per_group=M // G if G > 1 else 0))
    b = t(lambda: ops.skinny_tn(y, U, False, n_groups=G, lazy=True))
    c = t(lambda: ops.lora_side(y, U=U, Vt=Vt, n_groups=G))
    d = t(lambda: ops.lora_side(y, U=da, n_groups=2))
    e = t(lambda: ops.skinny_tn(y, da, True, n_groups=2, lazy=True))
    f = t(lambda: ops.lora_side(y, Vt=Vt, n_groups=G))
    print(f"C={C} G={G} ({mb:.1f} MB): skinny_nt {a:.1f} us + skinny_tn {b:.1f} us  ->  lora_side both {c:.1f} us ({mb / c:.2f} TB/s... MB/us) | "
          f"dA pass: skinny_tn {e:.1f} -> lora_side {d:.1f} us | P only {f:.1f} us", flush=True)
