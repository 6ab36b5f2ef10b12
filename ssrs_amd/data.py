"""Device-side training input pipeline: what `MultiScaleDataset.__getitem__` does to a
patch after it has been cut out (multiscale_dataset.py:190-215, 228-245) - random horizontal / vertical flip and
rot90, ImageNet normalisation of the uint8 RGB, per-patch DSM min-max - as two kernel launches on a whole batch
that is already on the GPU, so the loader workers only have to deliver raw uint8 / fp32 patches.

Not mirrored: the brightness / contrast jitter and the skimage rescale branch of `_apply_augmentation`
(:250-320; they need the host's skimage and per-channel means) and all file I/O."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # multiscale_dataset.py:37-38
IMAGENET_STD = (0.229, 0.224, 0.225)


def random_codes(n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Per-tile augmentation codes with the reference's probabilities (:228-245): hflip p = 0.5, vflip p = 0.5,
    with p = 0.5 a rotation by k in {1, 2, 3} quarter turns. int32 [n] on the CPU."""
    u = torch.rand((n, 3), generator=generator)
    k = torch.randint(1, 4, (n,), generator=generator)
    code = (u[:, 0] > 0.5).int() | ((u[:, 1] > 0.5).int() << 1) | (torch.where(u[:, 2] > 0.5, k, torch.zeros_like(k)).int() << 2)
    return code.to(torch.int32)


class DeviceAugment:
    """`aug(rgb_u8 [B,3,S,S] uint8, dsm [B,S,S] fp32, target [B,S,S] int64, codes=None)` -> `(rgb fp32 normalised,
    dsm min-max scaled per tile, target)`, all on the device and ready for `TrainEngine.step`."""

    def __init__(self, mean=IMAGENET_MEAN, std=IMAGENET_STD, augment: bool = True, generator: Optional[torch.Generator] = None):
        self.mean, self.std, self.augment, self.generator = tuple(mean), tuple(std), augment, generator

    def __call__(self, rgb_u8: torch.Tensor, dsm: torch.Tensor, target: Optional[torch.Tensor] = None,
                 codes: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        n = rgb_u8.shape[0]
        if codes is None:
            codes = random_codes(n, self.generator) if self.augment else torch.zeros(n, dtype=torch.int32)
        codes = codes.to(rgb_u8.device, non_blocking=True)
        rgb, d, t = ops.prepare_tiles(rgb_u8.contiguous(), dsm.float().contiguous(), None if target is None else target.contiguous(),
                                      codes, self.mean, self.std)
        ops.tile_minmax_norm(d, keep_flat=True)              # a flat patch stays as it is (:206)
        return rgb, d, t
