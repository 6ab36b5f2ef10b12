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
