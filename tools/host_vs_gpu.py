#!/usr/bin/env python3
"""Developer aid: host enqueue time vs GPU time of one training step (is the step launch-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssrs_amd import detfill
from ssrs_amd.trainer import TrainEngine
mod = sys.argv[1] if len(sys.argv) > 1 else "sam_lora"
dev = torch.device("cuda", 0)
m = bench.build_model(mod, "vit_b", 6, dev)
eng = TrainEngine(m, lr=0.01, class_weight=torch.ones(6, device=dev))
rgb, dsm, tgt = [t.to(dev) for t in detfill.tiles("bench/rank0", 16, 256, 6)]
for _ in range(3):
    eng.step(rgb, dsm, tgt)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.step(rgb, dsm, tgt)
    e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{mod}: host enqueue {1e3 * th:.2f} ms, GPU span {e0.elapsed_time(e1):.2f} ms, wall {1e3 * (time.perf_counter() - t0):.2f} ms")
