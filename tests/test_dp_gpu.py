"""Data-parallel gradient exchange of the REAL model with 2 ranks (trainer.py:146-197 sharded over ranks): two fresh processes
share the one GPU of the test box and exchange through gloo (staged over the host - RCCL needs one GPU per rank), walking
bank.deliver -> notify -> bucket all-reduce exactly as on an 8-GPU node. With BatchNorm in eval mode (no batch coupling) the
averaged gradients must equal the single-process gradients of the concatenated batch."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _tiny(train=False, mod="sam_lora"):
    import argparse
    import warnings
    from ssrs_amd import detfill
    from ssrs_amd.unetformer_mmsam import UNetFormer
    args = argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder="vit_tiny_test", args=args, dropout=0.0 if train else 0.1)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "dp/"))
    return m.cuda().train() if train else m.cuda().eval()


def _stage_collectives():
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    class Done:
        def wait(self):
            return True

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        h = t.detach().cpu()
        real_ar(h, op=op, group=group)
        t.copy_(h)
        return Done() if async_op else None

    def broadcast(t, src=0, group=None, async_op=False):
        h = t.detach().cpu()
        real_bc(h, src=src, group=group)
        t.copy_(h)
        return Done() if async_op else None
    dist.all_reduce, dist.broadcast = all_reduce, broadcast


def _worker(rank, world, port, q, mod):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _stage_collectives()
        from ssrs_amd import detfill, ops
        from ssrs_amd.trainer import TrainEngine, trainable_tensors
        rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("dp", 2, 256, 6)]
        w = torch.ones(6, device="cuda")
        # single-process gradients of the whole batch, plain autograd, no engine
        ref_m = _tiny(mod=mod)
        named, _ = trainable_tensors(ref_m)
        token = "lora_" if mod == "sam_lora" else "Adapter"   # the PEFT family of this variant is what sits in bucket 1
        assert any(token in n for n, _ in named) and not any(("Adapter" if token == "lora_" else "lora_") in n for n, _ in named)
        out = ref_m(rgb, dsm, mode="Train").contiguous()
        acc, _ = ops.ce_loss_fwd(out, tgt, w, -100)
        out.backward(ops.ce_loss_bwd(out, tgt, acc, w, -100, 1.0))
        ref = torch.cat([p.grad.reshape(-1) for _, p in named])
        worst = 0.0
        for overlap in (True, False):
            m = _tiny(mod=mod)
            if rank == 1:                                   # replicas that start apart must be pulled together by the engine
                with torch.no_grad():
                    m.decoder.segmentation_head[2][0].weight.add_(0.25)
            eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=w, overlap=overlap)
            assert eng.world == 2 and len(eng.bucket_bounds) == 2
            sl = slice(rank, rank + 1)
            eng.zero_grad()
            eng.forward_backward(rgb[sl].contiguous(), dsm[sl].contiguous(), tgt[sl].contiguous())
            if overlap:
                assert len(eng._works) == 2                  # both buckets went out from the gradient deliveries, during backward
            eng._finish_exchange()
            got = torch.cat([p.grad.reshape(-1) for p in eng.params]) / world
            worst = max(worst, ((got - ref).abs().max() / ref.abs().max()).item())
        q.put((rank, worst, None))
    except Exception as e:                                    # noqa: BLE001
        import traceback
        q.put((rank, float("nan"), traceback.format_exc()))
        raise e
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("mod,port", [("sam_lora", 29631), ("sam_adpt", 29633)])      # BASELINE configs[2] is the MMAdapter variant
def test_two_rank_gradients_equal_single_process_gradients(mod, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mod)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
    for rank, worst, err in res:
        assert err is None, err
        assert worst < 1e-5, (rank, worst)              # measured 2e-7 .. 3e-7
    assert all(p.exitcode == 0 for p in procs)


def _bn_buffers(m):
    return torch.cat([b.detach().float().reshape(-1) for n, b in m.named_buffers() if n.endswith(("running_mean", "running_var"))])


def _worker_sync_bn(rank, world, port, q, mod):
    """BatchNorm in TRAINING mode: with sync_bn the two ranks' one-tile batches must behave like the single-process two-tile
    batch (torch.nn.SyncBatchNorm's contract; SURVEY 8(e) strict 1-vs-N mode) - loss, averaged gradients, running statistics.
    Per-replica statistics (the default) are measured beside it: they must be far off, or the test proves nothing."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _stage_collectives()
        from ssrs_amd import detfill, ops
        from ssrs_amd.trainer import TrainEngine, trainable_tensors
        rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("dpbn", 2, 256, 6)]
        w = torch.ones(6, device="cuda")
        ref_m = _tiny(train=True, mod=mod)
        named, _ = trainable_tensors(ref_m)
        out = ref_m(rgb, dsm, mode="Train").contiguous()
        acc, _ = ops.ce_loss_fwd(out, tgt, w, -100)
        out.backward(ops.ce_loss_bwd(out, tgt, acc, w, -100, 1.0))
        ref = torch.cat([p.grad.reshape(-1) for _, p in named])
        ref_loss = (acc[0] / acc[1]).item()
        ref_buf = _bn_buffers(ref_m)
        res = {}
        for sync in (True, False):
            m = _tiny(train=True, mod=mod)
            eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=w, overlap=False, sync_bn=sync)
            sl = slice(rank, rank + 1)
            eng.zero_grad()
            a, _ = eng.forward_backward(rgb[sl].contiguous(), dsm[sl].contiguous(), tgt[sl].contiguous())
            eng._finish_exchange()
            a = a.clone()
            dist.all_reduce(a)
            got = torch.cat([p.grad.reshape(-1) for p in eng.params]) / world
            res[sync] = (((got - ref).norm() / ref.norm()).item(), abs((a[0] / a[1]).item() - ref_loss),
                         ((_bn_buffers(m) - ref_buf).abs().max() / ref_buf.abs().max()).item())
        q.put((rank, res, None))
    except Exception as e:                                    # noqa: BLE001
        import traceback
        q.put((rank, None, traceback.format_exc()))
        raise e
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("mod,port", [("sam_lora", 29641), ("sam_adpt", 29643)])
def test_sync_bn_two_ranks_equal_single_process_batch(mod, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sync_bn, args=(r, 2, port, q, mod)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
    for rank, r, err in res:
        assert err is None, err
        print(f"rank {rank}: sync_bn grad L2 {r[True][0]:.2e} loss {r[True][1]:.2e} running stats {r[True][2]:.2e} | "
              f"per-replica grad L2 {r[False][0]:.2e} loss {r[False][1]:.2e} running stats {r[False][2]:.2e}")
        # bf16 roundings carry fp32 summation-order differences of the statistics to bf16 level, and ReLU6 / max-pool switches
        # amplify that in the gradients: bars, not equality. Measured: sync 5.0e-2 / 7.9e-5 / 6.2e-4, per-replica
        # statistics 7.1e-1 / 3.0e-3 / 6.9e-2
        assert r[True][0] < 1e-1 and r[True][1] < 5e-4 and r[True][2] < 2e-3, (rank, r)
        assert r[False][0] > 5 * r[True][0], (rank, r)
    assert all(p.exitcode == 0 for p in procs)
