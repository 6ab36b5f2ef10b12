#!/usr/bin/env python3
"""Where a GEMM workgroup spends its time (diagnostic; needs tools/_diag/libgemm_stamps.so = gemm.hip built with -DGEMM_STAMPS):
per-workgroup real-time stamps (100 MHz) at entry / first tile landed / main loop done / stores issued / stores done.
    python tools/gemm_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssrs_amd import _lib  # noqa: E402

L = ctypes.CDLL(os.path.join(ROOT, "tools", "_diag", os.environ.get("STAMPS_LIB", "libgemm_stamps.so")))
L.mfnet_gemm_bf16_tiled.argtypes = _lib._SIGS["mfnet_gemm_bf16_tiled"]
L.mfnet_gemm_bf16_tiled.restype = ctypes.c_int
TILE = int(os.environ.get("STAMPS_TILE", "-1"))
L.mfnet_diag_set_stamps.argtypes = [ctypes.c_void_p]
L.mfnet_last_error.restype = ctypes.c_char_p
if hasattr(L, "mfnet_diag_set_stagger"):
    L.mfnet_diag_set_stagger.argtypes = [ctypes.c_int]
    L.mfnet_diag_set_stagger(int(os.environ.get("STAMPS_STAGGER", "0")))
TILE_M = int(os.environ.get("STAMPS_TILE_M", "128"))
TILE_N = int(os.environ.get("STAMPS_TILE_N", "128"))


def run(M, N, K, kind):
    dev = torch.device("cuda")
    x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    U = torch.rand(M, 8, device=dev)
    V = torch.rand(2, N, 8, device=dev)
    bias = torch.rand(N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    res = torch.rand(M, N, device=dev)
    out_bf = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out_f = torch.empty(M, N, device=dev)
    p = lambda t: None if t is None else t.data_ptr()      # noqa: E731
    cfg = dict(lin1=dict(bias=bias, lora=True, act=1, pre=pre, out=out_bf), lin2d=dict(lora=True, act=2, aux=aux, out=out_bf),
               qkv=dict(bias=bias, lora=True, out=out_bf), lin2=dict(bias=bias, lora=True, res=res, out=out_f),
               proj=dict(bias=bias, res=res, out=out_f), plain=dict(out=out_bf))[kind]
    o = cfg["out"]
    lora = cfg.get("lora", False)

    tile = TILE if TILE else {3072: 6, 768: 3, 2304: 9}[N]

    def launch():
        rc = L.mfnet_gemm_bf16_tiled(x.data_ptr(), K, w.data_ptr(), K, M, N, K, p(cfg.get("bias")), p(U) if lora else None, 8 if lora else 0,
                               p(V) if lora else None, V.stride(0) if lora else 0, M // 2 if lora else 0, 2 if lora else 1, 8 if lora else 0,
                               0.25, cfg.get("act", 0), p(cfg.get("aux")), N if cfg.get("aux") is not None else 0, p(cfg.get("pre")),
                               N if cfg.get("pre") is not None else 0, p(cfg.get("res")), N if cfg.get("res") is not None else 0, 0,
                               o.data_ptr(), 1 if o.dtype == torch.bfloat16 else 0, N, 1, tile, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, L.mfnet_last_error()
    tile = TILE if TILE else {3072: 6, 768: 3, 2304: 9}[N]
    nwg = (M // 128) * (N // 128) if tile < 0 else (M // 256) * (N // (32 * tile))
    st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    L.mfnet_diag_set_stamps(None)
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    L.mfnet_diag_set_stamps(st.data_ptr())
    launch()
    torch.cuda.synchronize()
    L.mfnet_diag_set_stamps(None)
    s = st.cpu().numpy().reshape(nwg, 8).astype(np.int64)
    if os.environ.get("STAMPS_DUMP"):
        np.save(os.path.join(ROOT, "gpurun_out", f"stamps_{kind}_{M}_{N}_{K}.npy"), s)
    t = (s[:, :5] - s[:, 0].min()) / 100.0             # us since the first workgroup started
    cu = (s[:, 6] & 0xf) * 100000 + (s[:, 5] & 0xffffff00 >> 0 & 0xfff00)   # (xcc, se/sh/cu bits of HW_ID)
    names = ["prologue (entry -> first tile landed)", "main loop", "epilogue until stores issued", "stores drain"]
    print(f"\n== {kind} M,N,K=({M},{N},{K})  {us:.1f} us/launch back-to-back, {2 * M * N * K / us / 1e6:.0f} TF/s; {nwg} workgroups, "
          f"kernel span by stamps {t[:, 4].max():.1f} us")
    for i, n in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"   {n:40s} median {np.median(d):6.2f}  p10 {np.quantile(d, .1):6.2f}  p90 {np.quantile(d, .9):6.2f} us")
    life = t[:, 4] - t[:, 0]
    print(f"   workgroup lifetime median {np.median(life):.2f} us; starts: p50 {np.median(t[:, 0]):.1f} p90 {np.quantile(t[:, 0], .9):.1f} max {t[:, 0].max():.1f} us")
    ncu = len(np.unique(cu))
    # how many workgroups are in their main loop / epilogue at a time (sampled every 0.5 us)
    grid = np.arange(0, t[:, 4].max(), 0.5)
    in_main = [(np.logical_and(t[:, 1] <= g, g < t[:, 2])).sum() for g in grid]
    in_epi = [(np.logical_and(t[:, 2] <= g, g < t[:, 4])).sum() for g in grid]
    in_pro = [(np.logical_and(t[:, 0] <= g, g < t[:, 1])).sum() for g in grid]
    print(f"   distinct CU ids seen {ncu}; time-average workgroups in prologue {np.mean(in_pro):.0f}, main loop {np.mean(in_main):.0f}, epilogue {np.mean(in_epi):.0f}")
    print("   timeline (every 4 us): " + "  ".join(f"{g:.0f}us:{a}/{b}/{c}" for g, a, b, c in list(zip(grid, in_pro, in_main, in_epi))[::8]))


if __name__ == "__main__":
    torch.manual_seed(0)
    only = os.environ.get("STAMPS_ONLY")
    for shp in [(8192, 3072, 768, "lin1"), (8192, 3072, 768, "lin2d"), (8192, 3072, 768, "plain"), (8192, 768, 3072, "lin2"), (8192, 768, 3072, "plain"),
                (8192, 768, 768, "proj"), (8192, 768, 768, "plain"), (8192, 2304, 768, "qkv"), (8192, 768, 2304, "plain")]:
        if only is None or shp[3] in only.split(","):
            run(*shp)
