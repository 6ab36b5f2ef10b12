"""Shared test helpers."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / max |b| (the 'relative' measure used for the 1e-4 fp32 / 2e-2 bf16 tolerances)."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf16r(t):
    """round to bf16 and back (what a bf16 kernel input actually holds)."""
    return t.to(torch.bfloat16).to(torch.float32)


def oracle_train_step_grads(tag, mod, dtype, ncls=6):
    """The tiny-model train step of tests/golden/<tag>.npz through the CPU oracle in `dtype` -> (loss, logits, {name: gradient}).
    In float64 this reproduces the reference's record to 1e-9 (test_oracle_golden.py); in float32 it shows what torch's own fp32
    arithmetic makes of the same step."""
    import argparse
    import warnings
    from oracle import mfnet_ref as R
    from ssrs_amd import detfill
    from ssrs_amd.unetformer_mmsam import UNetFormer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=ncls, encoder="vit_tiny_test", args=argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1))
    sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in detfill.fill_state_dict(m.state_dict(), tag + "/").items()}
    cfg = R.vit_cfg("tiny", mod=mod, dim=128, depth=2, heads=2, global_idx=(1,), num_classes=ncls)
    names = [str(n) for n in golden(tag)["grad_names"]]
    key = lambda n: n[4:] if n.startswith("sam.image_encoder.") else n      # noqa: E731
    params = {n: sd[key(n)].clone().requires_grad_() for n in names}
    rgb, dsm, tgt = detfill.tiles(tag, 2, 256, ncls)
    work = dict(sd)
    work.update({key(n): v for n, v in params.items()})
    out = R.unetformer(work, rgb.to(dtype), dsm.to(dtype), cfg, training=True, bn_state=R.BNState(True))
    loss = R.cross_entropy_2d(out, tgt, torch.ones(ncls, dtype=dtype))
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.item(), out.detach(), {n: g.detach() for n, g in zip(names, grads)}


def summary_err(got, want, floor=0.0):
    """(norm error, leading-slice error) of a tensor against a summarize() record [norm, sum, first 256] of the reference. floor:
    the magnitude below which a reference value counts as zero (a bias in front of a training-mode BatchNorm has an analytically
    zero gradient: the reference records 1e-17 of float64 noise there, fp32 arithmetic leaves 1e-9) - errors are relative to
    max(|reference|, floor)."""
    f = torch.as_tensor(got).detach().double().flatten().cpu()
    nerr = abs(f.norm().item() - want[0]) / max(want[0], floor, 1e-300)
    sl = f[:256].numpy()
    w = want[2:2 + sl.size]
    return nerr, float(np.abs(sl - w).max() / max(np.abs(w).max(), floor, 1e-300))
