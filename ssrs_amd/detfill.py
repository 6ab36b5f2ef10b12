"""Closed-form deterministic tensors (weights / inputs) that regenerate bit-identically anywhere.

Golden vectors under tests/golden/ hold only OUTPUTS; the weights and inputs that produced them
are rebuilt from this closed form, so neither the reference nor a weight
file has to travel to the GPU box. Values come from a splitmix64-style integer hash of
(crc32(key), element index) -> uniform [-1, 1), then a per-key affine rule chosen from the key
name / shape so that activations stay O(1) through the network and every zero-initialised tensor
of the reference (rel_h/rel_w, lora_B*) is exercised.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(key: str, shape: Iterable[int]) -> np.ndarray:
    """float64 array of the given shape, uniform in [-1, 1), a pure function of (key, index)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    seed = np.uint64(zlib.crc32(key.encode()))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (seed << np.uint64(32))
    h = _mix(_mix(idx))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (2.0 * u - 1.0).reshape(shape)


def fill_like(key: str, ref: torch.Tensor) -> torch.Tensor:
    """Deterministic value for state-dict entry ``key`` with the shape/dtype of ``ref``."""
    shape = tuple(ref.shape)
    leaf = key.rsplit(".", 1)[-1]
    if not ref.dtype.is_floating_point:
        if leaf == "num_batches_tracked":
            return torch.zeros_like(ref)
        return ref.clone()                                  # e.g. relative_position_index
    u = uniform(key, shape)
    if leaf == "running_var":
        v = 1.0 + 0.25 * u
    elif leaf == "running_mean":
        v = 0.1 * u
    elif leaf == "pos_embed":
        v = 0.1 * u
    elif leaf in ("rel_h", "rel_w"):
        v = 0.06 * u
    elif leaf == "relative_position_bias_table":
        v = 0.3 * u
    elif leaf in ("lora_wx", "lora_wy", "wx_Adapter", "wy_Adapter"):
        v = 0.5 + 0.2 * u
    elif leaf == "weights":                                  # WF / FRH fuse weights (2,)
        v = 1.0 + 0.3 * u
    elif leaf == "positional_encoding_gaussian_matrix":
        v = u
    elif len(shape) >= 2:
        fan_in = max(1, int(np.prod(shape[1:])))
        if leaf.startswith("lora_B"):
            v = 0.3 * u
        else:
            v = u * (1.7 / np.sqrt(fan_in))
    elif leaf == "weight":                                   # 1-D norm scales
        v = 1.0 + 0.1 * u
    else:                                                    # biases
        v = 0.05 * u
    return torch.from_numpy(np.ascontiguousarray(v)).to(ref.dtype)


def fill_state_dict(template: Dict[str, torch.Tensor], tag: str = "") -> Dict[str, torch.Tensor]:
    """New state dict with every entry of ``template`` deterministically filled. Keys that alias
    the same module under two names (``sam.image_encoder.*`` / ``image_encoder.*``) get identical
    values because the hash key strips a leading ``sam.``."""
    out = {}
    for k, v in template.items():
        hk = k[4:] if k.startswith("sam.image_encoder.") else k
        out[k] = fill_like(tag + hk, v)
    return out


def tiles(key: str, batch: int, size: int = 256, n_classes: int = 6) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Synthetic (rgb, dsm, target) batch: rgb ~ U[0,1) (B,3,S,S), dsm ~ U[0,1) (B,S,S), target
    int64 in [0, n_classes) with a coarse blocky structure so argmax/accuracy are meaningful."""
    rgb = torch.from_numpy(0.5 * (uniform(key + "/rgb", (batch, 3, size, size)) + 1.0)).float()
    dsm = torch.from_numpy(0.5 * (uniform(key + "/dsm", (batch, size, size)) + 1.0)).float()
    coarse = 0.5 * (uniform(key + "/tgt", (batch, size // 16, size // 16)) + 1.0)
    tgt = np.minimum((coarse * n_classes).astype(np.int64), n_classes - 1)
    tgt = np.repeat(np.repeat(tgt, 16, axis=1), 16, axis=2)
    return rgb, dsm, torch.from_numpy(tgt)
