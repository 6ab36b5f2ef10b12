#!/usr/bin/env python3
"""Yardstick only (nothing in the product calls a library GEMM): what the vendor's tuned bf16 GEMM (torch.mm -> hipBLASLt / Tensile
`Cijk_*` kernels) reaches on the encoder's token-matrix shapes, back to back on hot operands - to be read beside the `plain` rows of
`tools/gemm_bench.py` (mfnet_gemm_bf16 on the same shapes, same conditions).
    python tools/library_gemm_ref.py"""
import torch, time
dev = "cuda"
shapes = [(8192, 768, 768), (8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 768, 2304), (16384, 1024, 256)]
for M, N, K in shapes:
    a = (torch.rand(M, K, device=dev) - 0.5).bfloat16()
    w = (torch.rand(N, K, device=dev) - 0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.mm(a, w.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        torch.mm(a, w.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 50
    print(f"torch.mm (library GEMM) M,N,K=({M},{N},{K}): {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s", flush=True)
