#!/usr/bin/env python3
"""Sliding-window inference throughput on a synthetic raster resident in HBM.

    python tools/bench_infer.py [--size 8192] [--batch 32] [--encoder vit_b] [--mod sam_lora]

(i)  inference.py grid: 256-px windows, overlap 0.25 -> stride 192, per-tile DSM min-max, softmax[1] blend
     (RasterInference; reference inference.py:498-568,687-744);
(ii) evaluator grid: stride 64 with clamped tail positions, whole-raster DSM min-max, pred += logits, arg-max
     (SlidingWindowPredictor; reference evaluator.py:113-176, utils2.py:134-165).
Prints one JSON line per grid: window visits per second and seconds per raster."""
import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--encoder", default="vit_b")
    ap.add_argument("--mod", default="sam_lora")
    ap.add_argument("--skip-evaluator", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every forward eagerly instead of replaying a HIP graph per batch shape")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    a = ap.parse_args()
    from ssrs_amd import detfill
    from ssrs_amd.inference import RasterInference, SlidingWindowPredictor, get_sliding_windows, sliding_window
    from ssrs_amd.unetformer_mmsam import UNetFormer
    dev = torch.device("cuda")
    out = []
    for ncls, kind in ((2, "inference.py grid"), (6, "evaluator grid")):
        if kind.startswith("evaluator") and a.skip_evaluator:
            continue
        args = argparse.Namespace(mod=a.mod, mid_dim=None, image_size=1024, multimask_output=1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = UNetFormer(num_classes=ncls, encoder=a.encoder, args=args, precision=a.precision)
        m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "infer/"))
        m = m.to(dev).eval()
        g = torch.Generator(device="cpu").manual_seed(0)
        rgb = torch.randint(0, 256, (3, a.size, a.size), dtype=torch.uint8, generator=g).to(dev)
        dsm = (torch.rand(a.size, a.size, generator=g) * 50.0).to(dev)
        fwd = lambda x, y: m(x, y)                           # noqa: E731
        if kind.startswith("inference"):
            eng = RasterInference(fwd, batch_size=a.batch, graph=not a.no_graph)
            n = len(get_sliding_windows(a.size, a.size, (256, 256), eng.overlap))
            run = lambda: eng.predict(rgb, dsm)              # noqa: E731
            small = (rgb[:, :2048, :2048].contiguous(), dsm[:2048, :2048].contiguous())
            eng.predict(*small)                                  # eager warm-up + graph capture of the full and the tail batch
        else:
            eng = SlidingWindowPredictor(fwd, ncls, batch_size=a.batch, graph=not a.no_graph)
            img = rgb.float() / 255.0
            n = sum(1 for _ in sliding_window((a.size, a.size), 64, (256, 256)))
            run = lambda: eng.predict(img, dsm, 64)          # noqa: E731
            eng.predict(img[:, :1024, :1024].contiguous(), dsm[:1024, :1024].contiguous(), 64)
            eng.predict(img[:, :1024, :1024].contiguous(), dsm[:1024, :1024].contiguous(), 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the same job handed over in (pinned) host buffers: uint8 RGB + fp32 DSM in, uint8 mask out (never `value`)
        rgb_h, dsm_h = rgb.cpu().pin_memory(), dsm.cpu().pin_memory()
        mask_dev = res[1] if isinstance(res, tuple) else res
        mask_h = torch.empty(mask_dev.shape, dtype=mask_dev.dtype).pin_memory()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rgb2, dsm2 = rgb_h.to(dev, non_blocking=True), dsm_h.to(dev, non_blocking=True)
        if kind.startswith("inference"):
            r2 = eng.predict(rgb2, dsm2)
            mask_h.copy_(r2[1], non_blocking=True)
        else:
            r2 = eng.predict(rgb2.float() / 255.0, dsm2, 64)
            mask_h.copy_(r2, non_blocking=True)
        torch.cuda.synchronize()
        dt_pcie = time.perf_counter() - t1
        line = {"metric": "sliding-window inference window visits/sec", "grid": kind, "raster": f"{a.size}x{a.size}", "windows": n,
                "value": round(n / dt, 1), "unit": "windows/s", "seconds_per_raster": round(dt, 3),
                "pcie_inclusive_windows_per_s": round(n / dt_pcie, 1), "pcie_inclusive_seconds": round(dt_pcie, 3), "batch": a.batch,
                "model": f"{a.encoder} {a.mod}, {ncls} classes", "dtype": a.precision, "hip_graph": not a.no_graph, "data": "synthetic"}
        print(json.dumps(line), flush=True)
        out.append(line)
        del m, eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
