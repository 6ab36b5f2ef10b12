#!/usr/bin/env python3
"""One GEMM shape launched a few times (for rocprofv3 --pmc runs): python tools/micro/one_gemm.py M N K [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssrs_amd import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
for _ in range(reps):
    ops.gemm(x, w)
torch.cuda.synchronize()
