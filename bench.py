#!/usr/bin/env python3
"""MFNet hot-path benchmark: train tiles/s of ViT-B MMLoRA on synthetic 256x256 RGB+DSM tiles.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mod sam_lora|sam_adpt] [--no-cpu]

One "step" = one full training step (forward, cross-entropy, backward, SGD with momentum + weight
decay) of `UNetFormer` (SAM ViT-B encoder, MMLoRA) over one batch of B=16 tiles per GPU that is
already resident in HBM (BASELINE.json configs[1]). For N>1 the driver launches one rank per GPU
with torch.distributed.run; the batch is sharded (weak scaling, B per GPU) and gradients are
all-reduced over RCCL, overlapped with the encoder backward (ssrs_amd/trainer.py).

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline     - the dominant kernel (gemm_big_kernel, MFMA-bound): algorithmic FLOPs per launch
                 (2*M*N*K averaged over its launches of one step) / its average launch duration,
                 measured with HIP events on the launch stream in a second pass over the same steps;
                 roofline_all_gemm: the same over both GEMM kernels (the figure of the earlier rounds);
  roofline_hbm - the largest HBM-bound kernel (ln_bwd_kernel) against 8 TB/s;
  cpu_baseline - the CPU restatement (oracle/, kind "port") timed on this box's host cores on
                 BASELINE.json configs[0] (B=2 fwd+CE+bwd+SGD, a bounded sample).
Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0               # HBM3E spec (6.3 TB/s is what a float4 copy reaches), same guide
TRAIN_GF_PER_TILE = {"sam_lora": 191.0, "sam_adpt": 211.0}     # SURVEY.md 8(d), tools/flops.py


def build_model(mod, encoder, num_classes, device, precision="bf16"):
    from ssrs_amd import detfill
    from ssrs_amd.unetformer_mmsam import UNetFormer
    args = argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=num_classes, encoder=encoder, args=args, precision=precision)
    # random-init weights of the right architecture from the closed-form fill (no checkpoint offline)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "bench/"))
    return m.to(device).train()


def cpu_baseline(mod, encoder, num_classes, budget_s=25.0):
    """The oracle's train step (fp32, torch CPU ops, autograd) on configs[0]: B=2 tiles."""
    from oracle import mfnet_ref as R
    from ssrs_amd import detfill
    from ssrs_amd.trainer import trainable_tensors
    m = build_model(mod, encoder, num_classes, "cpu")
    named, _ = trainable_tensors(m)
    key = lambda n: n[4:] if n.startswith("sam.image_encoder.") else n      # noqa: E731
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    params = {key(n): sd[key(n)].clone().requires_grad_() for n, _ in named}
    cfg = R.vit_cfg(encoder, mod=mod, num_classes=num_classes)
    B = 2
    rgb, dsm, tgt = detfill.tiles("bench/cpu", B, 256, num_classes)
    bufs = {}
    w = torch.ones(num_classes)
    times = []
    t_begin = time.time()
    while True:
        t0 = time.time()
        work = dict(sd)
        work.update(params)
        st = R.BNState(True)
        out = R.unetformer(work, rgb, dsm, cfg, training=True, bn_state=st)
        loss = R.cross_entropy_2d(out, tgt, w)
        grads = torch.autograd.grad(loss, list(params.values()))
        with torch.no_grad():
            R.sgd_step(params, dict(zip(params.keys(), grads)), bufs, 0.01, 0.9, 5e-4)
        times.append(time.time() - t0)
        if len(times) >= 2 and (time.time() - t_begin > budget_s or len(times) >= 12):
            break
    steady = times[1:]                         # first step pays allocator / thread-pool warm-up
    med = sorted(steady)[len(steady) // 2]
    return {"value": round(B / med, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 train step, ViT-B {mod}, B={B} 256x256 tiles (BASELINE configs[0]); "
                      f"median of {len(steady)} steps after 1 warm-up, {med:.3f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU")
    ap.add_argument("--mod", default="sam_lora")
    ap.add_argument("--encoder", default="vit_b")
    ap.add_argument("--classes", type=int, default=6)
    ap.add_argument("--tile", type=int, default=256, help="tile edge in pixels (BASELINE configs[1]: 256; config 5: 1024)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a HIP graph")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"], help="fp32: the fp32 training mode (every contraction on the fp32 MFMA; a side "
                    "measurement - the headline metric is the bf16 step)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the with_h2d leg (steps fed from pinned host memory through the prefetcher)")
    a = ap.parse_args()
    if a.precision == "fp32":
        a.no_roofline = True             # the roofline legs trace the bf16 MFMA GEMM launches

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
            # started plainly with --gpus N: become the launcher. No GPU call has been made in this process, so N fresh children
            # (one rank per GPU, RCCL over xGMI) are started through torch.distributed.run and this process only relays their
            # output and exit code.
            import socket
            import subprocess
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.run(cmd).returncode)
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU: the MFNet hot path has no CPU fallback")
    # MFNET_BENCH_REHEARSAL (developer aid on a one-GPU box; the numbers of such a run mean nothing):
    #   gloo  - N ranks share the visible GPUs and exchange over gloo through host copies: walks the N>1 control flow
    #   nccl1 - one rank, RCCL initialised with world size 1, the engine still issues its all-reduces: walks the
    #           RCCL + HIP-graph sequence of a step on a real GPU
    rehearsal = os.environ.get("MFNET_BENCH_REHEARSAL", "")
    if rehearsal:
        import faulthandler
        faulthandler.dump_traceback_later(120, exit=True)         # a hang names its line instead of idling
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if rehearsal == "gloo":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        real_all_reduce = dist.all_reduce

        class _Done:
            def wait(self):
                return True

        def staged_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
            h = t.detach().cpu()
            real_all_reduce(h, op=op, group=group)
            t.copy_(h)
            return _Done() if async_op else None
        real_broadcast = dist.broadcast

        def staged_broadcast(t, src=0, group=None, async_op=False):
            h = t.detach().cpu()
            real_broadcast(h, src=src, group=group)
            t.copy_(h)
            return _Done() if async_op else None
        dist.all_reduce, dist.broadcast = staged_all_reduce, staged_broadcast
    elif rehearsal == "nccl1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from ssrs_amd import detfill, ops
    from ssrs_amd.trainer import TrainEngine
    model = build_model(a.mod, a.encoder, a.classes, dev, a.precision)
    eng = TrainEngine(model, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(a.classes, device=dev),
                      use_graph=not a.eager, exchange_when_alone=rehearsal == "nccl1")
    rgb, dsm, tgt = detfill.tiles(f"bench/rank{rank}", a.batch, a.tile, a.classes)
    rgb, dsm, tgt = rgb.to(dev), dsm.to(dev), tgt.to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(a.warmup, 3 if eng.use_graph else 0)):      # graph mode: 2 eager steps + the capture are never timed
        acc = eng.step(rgb, dsm, tgt)
    sync()
    exchanging = dist.is_initialized() and (world > 1 or rehearsal == "nccl1")
    if exchanging:
        eng.exchange_events = []            # one event pair per step around the gradient collectives (2 event records: no sync)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        acc = eng.step(rgb, dsm, tgt)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    metrics = TrainEngine.read_metrics(acc)
    comm = None
    if exchanging:
        # makes the N > 1 record self-proving: which backend exchanged, over how many ranks, how many bytes, how long per step
        evs, eng.exchange_events = eng.exchange_events, None
        torch.cuda.synchronize()
        ar_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, len(evs))
        comm = {"backend": dist.get_backend(), "rccl_world": dist.get_world_size(), "grad_bytes": eng.grad_bytes,
                "allreduce_ms_per_step": round(ar_ms, 4), "allreduces_timed": len(evs),
                "exchange": "one all-reduce of the flat gradient buffer after the graph replay" if eng.use_graph
                            else "two buckets launched from the backward pass (overlapped)"}

    with_h2d = None
    if not a.no_h2d and a.tile == 256:
        # The same step fed the way a loader feeds it: raw uint8 RGB + fp32 DSM + int64 labels in HOST memory, staged through
        # pinned buffers and copied one batch ahead on a side stream (ssrs_amd.data.Prefetcher), augmented / normalised on the
        # device (DeviceAugment: flips, rot90, brightness / contrast jitter, ImageNet statistics, per-tile DSM min-max). A
        # reported leg beside `value`, whose timed region starts with the batch resident in HBM.
        from ssrs_amd.data import DeviceAugment, Prefetcher
        hsteps = min(a.steps, 50)
        g = torch.Generator().manual_seed(1234 + rank)
        host_rgb = torch.randint(0, 256, (a.batch, 3, a.tile, a.tile), dtype=torch.uint8, generator=g)
        host_dsm = 100 + 20 * torch.rand((a.batch, a.tile, a.tile), generator=g)
        host_tgt = tgt.cpu()
        aug = DeviceAugment(generator=g)
        feed = Prefetcher(((host_rgb, host_dsm, host_tgt) for _ in range(hsteps + 3)), device=dev, transform=aug)
        itr = iter(feed)
        for _ in range(3):
            eng.step(*next(itr))
        sync()
        th = time.perf_counter()
        for batch in itr:
            eng.step(*batch)
        sync()
        dth = time.perf_counter() - th
        if world > 1:
            t = torch.tensor([dth], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dth = t.item()
        nbytes = host_rgb.numel() + host_dsm.numel() * 4 + host_tgt.numel() * 8
        with_h2d = {"ms_per_step": round(1e3 * dth / hsteps, 3), "tiles_per_s": round(world * a.batch * hsteps / dth, 2), "steps": hsteps,
                    "h2d_bytes_per_step_per_gpu": int(nbytes),
                    "pipeline": "pinned double-buffered prefetch (side stream) + device flip/rot90/jitter/normalise/DSM min-max + train step"}

    roofline = roofline_all = roofline_hbm = roofline_attn = roofline_hbm_table = None
    if not a.no_roofline:
        # second pass over the same steps, eagerly, with an event pair around every GEMM launch. Every rank takes
        # it (the steps contain the gradient all-reduce, a collective); rank 0's launches are the ones reported.
        eng.use_graph = False
        ops.GEMM_TRACE, ops.LN_BWD_TRACE = [], []
        rsteps = min(a.steps, 20)            # (event pairs of 20 steps: ~3600 GEMM launches)
        for _ in range(rsteps):
            eng.step(rgb, dsm, tgt)
        torch.cuda.synchronize()
        tr, ops.GEMM_TRACE = ops.GEMM_TRACE, None
        tr_ln, ops.LN_BWD_TRACE = ops.LN_BWD_TRACE, None
        # third pass, a few steps: every other traced entry point (the HBM-bound kernels and attention), see ops.tcall
        ops.KERNEL_TRACE = {}
        ksteps = min(a.steps, 5)
        for _ in range(ksteps):
            eng.step(rgb, dsm, tgt)
        torch.cuda.synchronize()
        ktr, ops.KERNEL_TRACE = ops.KERNEL_TRACE, None
    if rank == 0 and not a.no_roofline:
        # an event pair with nothing between its two records still reads a few microseconds apart; that offset
        # (median of 64 empty pairs) is taken off every launch so the figure agrees with rocprofv3's kernel durations
        empty = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            empty.append((e0, e1))
        torch.cuda.synchronize()
        ev_over = sorted(a.elapsed_time(b) for a, b in empty)[32]
        # The step's dense products run in two kernels: gemm_big_kernel (256-row tiles: the encoder's token matrices - the dominant
        # kernel, ~4.1 of ~11.3 ms of kernel time) and gemm_bf16_kernel (128x128 tiles: ~60 small decoder / FPN launches that are
        # latency-bound at 5-25 us). `roofline` is the dominant kernel alone, as rocprofv3 lists it; `roofline_all_gemm` keeps the
        # figure of the earlier rounds (both kernels together). Which kernel a shape takes: csrc/gemm_big.hip big_choose_nf.
        def takes_big(shp):
            M, N, K = shp
            if M % 256 or K % 64:
                return False
            return any(N % (32 * nf) == 0 and (M // 256) * (N // (32 * nf)) >= 192 for nf in (9, 8, 6, 4, 3))

        def summarise(entries, kernel, traffic):
            ms = sum(max(t[0].elapsed_time(t[1]) - ev_over, 0.0) for t in entries)
            fl = sum(t[2] for t in entries)
            achieved = fl / (ms * 1e-3) / 1e12
            return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "launches_per_step": len(entries) // rsteps,
                    "avg_launch_us": round(1e3 * ms / len(entries), 2), "gflop_per_launch": round(fl / len(entries) / 1e9, 3),
                    "gemm_ms_per_step": round(ms / rsteps, 3), "event_pair_offset_us": round(1e3 * ev_over, 2)}

        # HBM bytes per launch come from the PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read from inside the
        # process): the newest profiles/*_pmc.json of each kernel, produced by tools/pmc_traffic.py from two rocprofv3 runs
        import glob

        traffic_files = []

        def stored_traffic(pattern):
            pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
            if pmc and a.mod == "sam_lora" and a.batch == 16 and a.encoder == "vit_b" and a.tile == 256:
                with open(pmc[-1]) as f:
                    traffic_files.append("profiles/" + os.path.basename(pmc[-1]))
                    return round(json.load(f)["traffic_bytes_per_launch"])
            return None
        tr_big = [t for t in tr if takes_big(t[3])]
        roofline_all = summarise(tr, "gemm_big_kernel + gemm_bf16_kernel", stored_traffic("*_gemm_pmc.json"))
        roofline = summarise(tr_big, "gemm_big_kernel", stored_traffic("*gemm_big_pmc.json")) if tr_big else roofline_all
        # the largest HBM-bound kernel of the step (LayerNorm backward of the residual stream: one streaming pass)
        ms_ln = sum(max(t[0].elapsed_time(t[1]) - ev_over, 0.0) for t in tr_ln)
        by_ln = sum(t[2] for t in tr_ln)
        if tr_ln and ms_ln > 0:
            gbs = by_ln / (ms_ln * 1e-3) / 1e9
            traffic_ln = stored_traffic("*ln_bwd_pmc.json")
            roofline_hbm = {"bound": "hbm", "kernel": "ln_bwd_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic_ln, "launches_per_step": len(tr_ln) // rsteps,
                            "avg_launch_us": round(1e3 * ms_ln / len(tr_ln), 2), "mbytes_per_launch": round(by_ln / len(tr_ln) / 1e6, 2)}
        # attention (MFMA-bound by its arithmetic, issue-bound in practice) and the ten largest HBM-bound kernels of the step:
        # algorithmic FLOPs / bytes (each operand once) over live event-pair durations
        def kagg(name):
            ent = ktr.get(name, [])
            ms = sum(max(e0.elapsed_time(e1) - ev_over, 0.0) for e0, e1, _, _ in ent)
            return ent, ms, sum(t[2] for t in ent), sum(t[3] for t in ent)
        roofline_attn = {}
        for nm in ("mfnet_attn_fwd", "mfnet_attn_bwd"):
            ent, ms, _, fl = kagg(nm)
            if ent and ms > 0:
                tf = fl / (ms * 1e-3) / 1e12
                roofline_attn[nm] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "launches_per_step": len(ent) // ksteps,
                                     "avg_launch_us": round(1e3 * ms / len(ent), 2), "gflop_per_launch": round(fl / len(ent) / 1e9, 3),
                                     "ms_per_step": round(ms / ksteps, 3)}
        hbm_rows = []
        for nm in ktr:
            ent, ms, by, fl = kagg(nm)
            if nm.startswith("mfnet_attn") or not ent or ms <= 0 or by <= 0:
                continue
            gbs = by / (ms * 1e-3) / 1e9
            hbm_rows.append({"kernel": nm, "launches_per_step": round(len(ent) / ksteps, 1), "avg_launch_us": round(1e3 * ms / len(ent), 2),
                             "mbytes_per_launch": round(by / len(ent) / 1e6, 3), "achieved": round(gbs, 1), "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_step": round(ms / ksteps, 4)})
        hbm_rows.sort(key=lambda r: -r["ms_per_step"])
        # the twelve largest, plus the kernels north_star names whatever their rank: patch embedding gather, decoder convolutions,
        # up-sampling, loss and optimizer
        named = ("mfnet_im2col_patch", "mfnet_up_final_fwd", "mfnet_up_final_bwd", "mfnet_upfuse_fwd", "mfnet_upfuse_bwd", "mfnet_dwconv_fwd",
                 "mfnet_dwconv_bwd_data", "mfnet_ce_loss_fwd", "mfnet_ce_loss_bwd", "mfnet_sgd_flat")
        hbm_rows = hbm_rows[:12] + [r for r in hbm_rows[12:] if r["kernel"] in named]
        roofline_hbm_table = {"bound": "hbm", "peak": HBM_PEAK_GBS, "rows": hbm_rows,
                              "note": "algorithmic bytes (each operand once) / live event time, eager pass; small maps are launch-latency bound"}
        if os.environ.get("MFNET_GEMM_SHAPES"):          # developer aid: per-shape GEMM time on stderr
            agg = {}
            for e0, e1, f, shp in tr:
                d = agg.setdefault(shp, [0, 0.0, f])
                d[0] += 1; d[1] += max(e0.elapsed_time(e1) - ev_over, 0.0)
            for shp, (n, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
                print(f"gemm M,N,K={shp}: {n // rsteps}/step {1e3 * t / n:8.1f} us  {f / (1e9 * t / n):8.1f} TF/s  {t / rsteps:7.3f} ms/step", file=sys.stderr)
    if world > 1:
        dist.barrier()

    if rank == 0:
        tiles = world * a.batch * a.steps
        base_cfg = a.mod == "sam_lora" and a.encoder == "vit_b" and a.tile == 256 and a.precision == "bf16"
        out = {"metric": "train tiles/sec (256x256 RGB+DSM, ViT-B MMLoRA)" if base_cfg else f"train tiles/sec ({a.tile}x{a.tile} RGB+DSM, {a.encoder} {a.mod})",
               "value": round(tiles / dt, 2), "unit": "tiles/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.precision, "data": "synthetic",
               "config": {"workload": f"{a.encoder} {a.mod} fine-tune step (fwd+CE+bwd+SGD), {a.batch} tiles of {a.tile}x{a.tile} RGB+DSM per GPU, "
                                      f"{a.classes} classes, " + ("bf16 MFMA / fp32 accumulate+master" if a.precision == "bf16" else "fp32 MFMA, fp32 activations and gradients") + (" (BASELINE configs[1])" if base_cfg and a.batch == 16 else ""),
                          "tiles_per_gpu": a.batch, "global_batch": world * a.batch, "parallelism": f"dp{world}"},
               "loss": round(metrics["loss"], 5),
               "algorithmic_train_tflops": (round(world * a.batch * TRAIN_GF_PER_TILE.get(a.mod, 191.0) * 1e9 / (dt / a.steps) / 1e12, 2)
                                            if a.encoder == "vit_b" and a.tile == 256 else None)}
        if roofline is not None:
            out["roofline"] = roofline
            out["roofline_all_gemm"] = roofline_all
            # `traffic` cannot be read from inside the process: it is the PMC measurement of this build's last profiled run
            out["traffic_source"] = {"files": traffic_files, "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/measure_round.sh), per launch"}
        if roofline_hbm is not None:
            out["roofline_hbm"] = roofline_hbm
        if not a.no_roofline:
            out["roofline_attn"] = roofline_attn
            out["roofline_hbm_table"] = roofline_hbm_table
        if with_h2d is not None:
            out["with_h2d"] = with_h2d
        if comm is not None:
            out["comm"] = comm
        if rehearsal:
            out["rehearsal"] = rehearsal + ": control-flow check only, not a measurement"
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(a.mod, a.encoder, a.classes)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
