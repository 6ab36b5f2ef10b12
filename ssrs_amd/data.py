"""Device-side training input pipeline: what `MultiScaleDataset.__getitem__` does to a
patch after it has been cut out (multiscale_dataset.py:190-215, 228-265) - random horizontal / vertical flip and
rot90, the brightness / contrast jitter, ImageNet normalisation of the uint8 RGB, per-patch DSM min-max - as three kernel
launches on a whole batch that is already on the GPU, so the loader workers only have to deliver raw uint8 / fp32 patches;
and `Prefetcher`, which brings those raw batches over PCIe from pinned host memory on a side stream, one batch ahead of
the step that consumes them.

Not mirrored: the skimage rescale branch of `_apply_augmentation` (:267-320; `skimage.transform.resize` with
anti-aliasing is a host library that is absent here) and all file I/O."""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Tuple

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


def random_jitter(n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Per-tile (brightness, contrast) factors with the reference's distribution (:250-252): with p = 0.7 both uniform in
    [0.8, 1.2], else (0, 0) = no jitter. fp32 [n, 2] on the CPU."""
    u = torch.rand((n, 3), generator=generator)
    f = 0.8 + 0.4 * u[:, 1:]
    return torch.where((u[:, :1] > 0.3), f, torch.zeros_like(f)).to(torch.float32).contiguous()


class DeviceAugment:
    """`aug(rgb_u8 [B,3,S,S] uint8, dsm [B,S,S] fp32, target [B,S,S] int64, codes=None, jitter=None)` -> `(rgb fp32
    normalised, dsm min-max scaled per tile, target)`, all on the device and ready for `TrainEngine.step`.
    codes: int32 [B] flip / rot90 codes (`random_codes`); jitter: fp32 [B,2] (brightness, contrast) per tile, a
    brightness <= 0 meaning none (`random_jitter`). With augment=True missing ones are drawn with the reference's
    probabilities. The jitter acts on the uint8 patch before the normalisation, as in the reference; the input batch is
    not modified (the jittered copy is a device temporary)."""

    def __init__(self, mean=IMAGENET_MEAN, std=IMAGENET_STD, augment: bool = True, generator: Optional[torch.Generator] = None,
                 jitter: bool = True):
        self.mean, self.std, self.augment, self.generator, self.jitter = tuple(mean), tuple(std), augment, generator, jitter

    def __call__(self, rgb_u8: torch.Tensor, dsm: torch.Tensor, target: Optional[torch.Tensor] = None,
                 codes: Optional[torch.Tensor] = None, jitter: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        n = rgb_u8.shape[0]
        if codes is None:           # nothing scripted by the caller: draw the augmentation (explicit codes without jitter = no jitter)
            codes = random_codes(n, self.generator) if self.augment else torch.zeros(n, dtype=torch.int32)
            if jitter is None and self.augment and self.jitter:
                jitter = random_jitter(n, self.generator)
        codes = codes.to(rgb_u8.device, non_blocking=True)
        rgb_u8 = rgb_u8.contiguous()
        if jitter is not None:
            # flips / rotations permute the pixels of a channel and the jitter is pointwise around the channel mean, so the
            # reference's order (flip, rotate, then jitter) and this one (jitter, then the fused flip + normalise) agree
            rgb_u8 = ops.jitter_tiles(rgb_u8.clone(), jitter.to(rgb_u8.device, non_blocking=True).float().contiguous())
        rgb, d, t = ops.prepare_tiles(rgb_u8, dsm.float().contiguous(), None if target is None else target.contiguous(),
                                      codes, self.mean, self.std)
        ops.tile_minmax_norm(d, keep_flat=True)              # a flat patch stays as it is (:206)
        return rgb, d, t


class Prefetcher:
    """Double-buffered host -> device feeder for the train loop: wraps any iterable of `(rgb, dsm, target)` CPU batches (the
    reference's DataLoader output, train_multiscale_main.py:507-522; raw uint8 RGB for `DeviceAugment`, or fp32) and yields them
    on the device. Each batch is staged into PINNED host buffers (allocated once per shape, two sets) and copied on a side
    stream while the previous batch's step runs; the consumer's stream waits on the copy's event, the host never blocks on the
    GPU. `transform` (e.g. a DeviceAugment) runs on the consumer's stream after that wait."""

    def __init__(self, loader: Iterable, device="cuda", transform=None, depth: int = 2):
        self.loader, self.device, self.transform, self.depth = loader, torch.device(device), transform, max(2, int(depth))
        self._stream = torch.cuda.Stream(device=self.device)
        self._pinned = {}

    def _stage(self, slot: int, k: int, t: torch.Tensor) -> torch.Tensor:
        key = (slot, k, tuple(t.shape), t.dtype)
        buf = self._pinned.get(key)
        if buf is None:
            buf = self._pinned[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        buf.copy_(t)
        return buf

    def _issue(self, slot: int, batch, done_with_slot: Optional[torch.cuda.Event]):
        if done_with_slot is not None:
            done_with_slot.synchronize()           # the pinned buffers of this slot are free once their previous copy has run
        host = [self._stage(slot, k, torch.as_tensor(t)) for k, t in enumerate(batch)]
        with torch.cuda.stream(self._stream):
            dev = [h.to(self.device, non_blocking=True) for h in host]
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return dev, ev

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, ...]]:
        it = iter(self.loader)
        pending, copied = [], [None] * self.depth
        slot = 0
        for batch in it:
            dev, ev = self._issue(slot, batch, copied[slot])
            copied[slot] = ev
            pending.append((dev, ev))
            slot = (slot + 1) % self.depth
            if len(pending) >= self.depth:
                yield self._hand_over(*pending.pop(0))
        while pending:
            yield self._hand_over(*pending.pop(0))

    def _hand_over(self, dev, ev):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in dev:
            t.record_stream(cur)                   # allocated on the side stream, used on the consumer's
        return self.transform(*dev) if self.transform is not None else tuple(dev)
