"""Sliding-window large-raster prediction for MFNet with the tile loop, normalisation, overlap merge and
arg-max on the device, and tiles sharded over the ranks of a torch.distributed group.

Two reference call paths are covered (paths relative to /root/reference):
  * `Evaluator._predict_image` (evaluator.py:113-176) with `utils2.sliding_window / grouper`
    (utils2.py:134-165): windows at range(0, H, stride) with the tail clamped to H - win, DSM min-max over
    the WHOLE raster, raw logits SUMMED over overlapping windows, arg-max  -> `SlidingWindowPredictor`.
  * `BuildingInference` tile loop (inference.py:498-568, 687-744): stride = win - int(win*overlap), tail
    tile shifted to the edge, short tiles reflect-padded, ImageNet-normalised RGB, PER-TILE DSM min-max,
    softmax probability of class 1 -> `RasterInference`. The reference vectorises every tile on the CPU
    and merges polygons (out of scope); here the per-tile probabilities are blended on the
    device (mean over the covering tiles) into one raster + a thresholded mask that a vectoriser can take.

Multi-GPU (one process per GPU): the window list is cut into `world` contiguous runs (row-major, so each
rank works on a horizontal band of the raster and keeps an accumulator for that band only); rows that
two neighbouring bands share are passed down a chain of point-to-point transfers (rank r -> r+1), then
every rank arg-maxes the rows it owns and the uint8 masks are combined. No collective touches logits.
"""
from __future__ import annotations

from itertools import islice
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .bank import EvalBank


# ---------------------------------------------------------------------------------------------- window grids (host)
def sliding_window(top, step=10, window_size=(20, 20)):
    """utils2.py:134-142: yields (x, y, w, h) with x the ROW offset; tail positions clamped to the edge."""
    H, W = top.shape[:2] if hasattr(top, "shape") else top
    for x in range(0, H, step):
        if x + window_size[0] > H:
            x = H - window_size[0]
        for y in range(0, W, step):
            if y + window_size[1] > W:
                y = W - window_size[1]
            yield x, y, window_size[0], window_size[1]


def count_sliding_window(top, step=10, window_size=(20, 20)):
    """utils2.py:145-155."""
    return sum(1 for _ in sliding_window(top, step, window_size))


def grouper(n, iterable):
    """utils2.py:158-165: chunks of n items (the last one shorter)."""
    it = iter(iterable)
    while True:
        chunk = tuple(islice(it, n))
        if not chunk:
            return
        yield chunk


def get_sliding_windows(height: int, width: int, window_size=(256, 256), overlap: int = 64):
    """inference.py:498-533 (_get_sliding_windows): (y, x, y_end, x_end); `overlap` in pixels."""
    wh, ww = window_size
    sh, sw = wh - overlap, ww - overlap
    out = []
    for y in range(0, height, sh):
        for x in range(0, width, sw):
            ye, xe = min(y + wh, height), min(x + ww, width)
            if (ye - y) >= wh // 2 and (xe - x) >= ww // 2:
                yy, xx = y, x
                if ye == height and ye - yy < wh:
                    yy = max(0, height - wh)
                if xe == width and xe - xx < ww:
                    xx = max(0, width - ww)
                out.append((yy, xx, ye, xe))
    return out


# ---------------------------------------------------------------------------------------------- band plan (host)
def plan_bands(rows: Sequence[int], heights: Sequence[int], H: int, world: int):
    """Cut a row-major window list (top rows `rows`, non-decreasing; tile heights `heights`) into `world`
    contiguous runs. Returns one dict per rank: first/last window index (half-open), band rows [lo, hi)
    covered by its windows, owned rows [own_lo, own_hi) (a partition of [0, H)). Rank r sends the rows
    [own_hi, hi) of its accumulator to rank r+1 after having received from rank r-1."""
    n = len(rows)
    assert n > 0 and all(rows[i] <= rows[i + 1] for i in range(n - 1)), "window rows must be non-decreasing"
    world = max(1, min(world, n))
    cuts = [round(i * n / world) for i in range(world + 1)]
    plan = []
    for r in range(world):
        a, b = cuts[r], cuts[r + 1]
        lo = rows[a]
        hi = max(rows[i] + heights[i] for i in range(a, b))
        plan.append(dict(rank=r, first=a, last=b, lo=lo, hi=hi))
    for r in range(world):
        plan[r]["own_lo"] = 0 if r == 0 else plan[r]["lo"]
        plan[r]["own_hi"] = H if r == world - 1 else plan[r + 1]["lo"]
        plan[r]["lo"] = plan[r]["own_lo"] if r == 0 else plan[r]["lo"]
        plan[r]["hi"] = max(plan[r]["hi"], plan[r]["own_hi"]) if r == world - 1 else plan[r]["hi"]
    for r in range(world - 1):                  # forwarded rows must fit inside the next band
        plan[r + 1]["hi"] = max(plan[r + 1]["hi"], plan[r]["hi"])
    return plan


def merge_bands(band: torch.Tensor, plan, rank: int, add_rows: Callable[[torch.Tensor, torch.Tensor], None], group=None):
    """Exchange of the rows neighbouring bands share. band: [C, hi-lo, W] accumulator of `rank`.
    add_rows(band, src) adds a received [C, k, W] block onto the top k rows of the band. Every rank posts its send (its rows
    below own_hi go to the next rank) and its receive (the previous rank's overhang) together, so the exchange is one
    neighbour round trip whatever the world size - the rows a rank sends are not the rows it receives into (a blocking
    receive-then-send made a chain whose latency grew with the number of ranks)."""
    world = len(plan)
    me = plan[rank]
    # bands thinner than a window: a rank's overhang reaches past its neighbour's own rows and must be FORWARDED after it was
    # received - only then is the exchange a chain (every rank sees the whole plan, so all take the same branch)
    if any(plan[r - 1]["hi"] > plan[r]["own_hi"] for r in range(1, world - 1)):
        if rank > 0:
            prev = plan[rank - 1]
            k = prev["hi"] - prev["own_hi"]
            if k > 0:
                buf = torch.empty((band.shape[0], k, band.shape[2]), dtype=band.dtype, device=band.device)
                dist.recv(buf, src=dist.get_global_rank(group, rank - 1) if group is not None else rank - 1, group=group)
                add_rows(band, buf)
        if rank < world - 1:
            k = me["hi"] - me["own_hi"]
            if k > 0:
                dist.send(band[:, me["own_hi"] - me["lo"]:me["hi"] - me["lo"], :].contiguous(),
                          dst=dist.get_global_rank(group, rank + 1) if group is not None else rank + 1, group=group)
        return
    ops_, buf = [], None
    if rank < world - 1:
        k = me["hi"] - me["own_hi"]
        if k > 0:
            out = band[:, me["own_hi"] - me["lo"]:me["hi"] - me["lo"], :].contiguous()
            dst = dist.get_global_rank(group, rank + 1) if group is not None else rank + 1
            ops_.append(dist.P2POp(dist.isend, out, dst, group))
    if rank > 0:
        prev = plan[rank - 1]
        k = prev["hi"] - prev["own_hi"]
        if k > 0:
            buf = torch.empty((band.shape[0], k, band.shape[2]), dtype=band.dtype, device=band.device)
            src = dist.get_global_rank(group, rank - 1) if group is not None else rank - 1
            ops_.append(dist.P2POp(dist.irecv, buf, src, group))
    if ops_:
        for req in dist.batch_isend_irecv(ops_):
            req.wait()
    if buf is not None:
        add_rows(band, buf)


def _coords4(tuples, H, W, h, w, device):
    """[(row, col, valid_h, valid_w)] -> int32 device tensor, after checking on the HOST that every window stays inside
    the raster and its valid extent inside the tile (the extraction kernel trusts these numbers)."""
    for r, c, vh, vw in tuples:
        if not (0 <= r and 0 <= c and 1 <= vh <= h and 1 <= vw <= w and r + vh <= H and c + vw <= W):
            raise ValueError(f"window (row {r}, col {c}, {vh}x{vw}) leaves the {H}x{W} raster or its {h}x{w} tile")
    if not tuples:
        return torch.zeros((0, 4), dtype=torch.int32, device=device)
    return torch.tensor(list(tuples), dtype=torch.int32, device=device)


def _dist_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _share_owned_rows(plan, group, *rasters):
    """Every rank ends up with the whole [H, W] rasters: rank r broadcasts the rows it owns (a partition of [0, H)), in
    place. Each row crosses the fabric once per receiver - an all-gather of the owned bands - where an all-reduce of the
    zero-padded full rasters moved (and summed) world times as much."""
    for r, pl in enumerate(plan):
        if pl["own_hi"] <= pl["own_lo"]:
            continue
        src = dist.get_global_rank(group, r) if group is not None else r
        for t in rasters:
            dist.broadcast(t[pl["own_lo"]:pl["own_hi"]], src=src, group=group)


def _device_add_rows(band, buf):
    """band[:, :k, :] += buf through the tile-accumulate kernel (one 'tile' of k x W at the origin)."""
    coords = torch.zeros((1, 2), dtype=torch.int32, device=band.device)
    ops.tile_accumulate(band, buf[None].contiguous(), coords, atomic=False)


def _int_add_rows(band, buf):
    """band[:, :k, :] += buf for the int64 blend accumulators (an exact integer add: nothing to round, nothing to order)."""
    band[:, :buf.shape[1], :] += buf


class GraphedForward:
    """`forward(rgb_tiles, dsm_tiles) -> logits` replayed from a HIP graph, one graph per input shape. The eval forward of
    MFNet is ~450 kernel launches of a few microseconds each for a 32-tile batch: enqueued eagerly from Python the host is
    the bottleneck. The first `warmup` calls of a shape run eagerly (they build the frozen-weight caches and set kernel
    attributes, which must not happen under capture); the next one captures. The returned tensor is the graph's static
    output: it is overwritten by the next call of the same shape, so consume it first (the predictors below do)."""

    def __init__(self, forward: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], warmup: int = 1, model=None):
        """model: the nn.Module(s) whose parameters the forward reads; found from the callable itself (a bound method, or a
        closure over a module / an object with a `.model`) when not given. Their parameter and buffer ADDRESSES are part of
        what a captured graph froze, so they are fingerprinted before every replay: a re-pointed parameter (`p.data = new`,
        `model.to(...)`, a TrainEngine adopting the model into its flat buffer) drops the graphs and the operand store instead
        of replaying against the old addresses."""
        self.forward, self.warmup = forward, warmup
        self._graphs, self._seen = {}, {}
        self._modules = _modules_of(forward) if model is None else ([model] if isinstance(model, torch.nn.Module) else list(model))
        self._print = None

    def _fingerprint(self):
        fp = []
        for m in self._modules:
            fp += [t.data_ptr() for t in m.parameters()]
            fp += [t.data_ptr() for t in m.buffers()]
        return tuple(fp)

    @torch.no_grad()
    def __call__(self, rgb: torch.Tensor, dsm: torch.Tensor) -> torch.Tensor:
        key = (tuple(rgb.shape), tuple(dsm.shape), rgb.dtype, dsm.dtype, rgb.device)
        if self._graphs and self._modules:
            fp = self._fingerprint()
            if fp != self._print:             # storage moved since the capture: nothing captured or stored may be used again
                self._graphs.clear()
                self._seen.clear()
                if EvalBank.active is not None:
                    EvalBank.active.entries.clear()
                    EvalBank.active.bn.clear()
        ent = self._graphs.get(key)
        if ent is None:
            n = self._seen.get(key, 0)
            if n < self.warmup:
                self._seen[key] = n + 1
                return self.forward(rgb, dsm)
            srgb, sdsm = rgb.clone(), dsm.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self.forward(srgb, sdsm)
            ent = self._graphs[key] = (g, srgb, sdsm, out)
            self._print = self._fingerprint() if self._modules else None
        g, srgb, sdsm, out = ent
        if EvalBank.active is not None:
            EvalBank.active.validate()        # the graph reads the stored weight-derived operands: refreshed in place if stale
        srgb.copy_(rgb, non_blocking=True)
        sdsm.copy_(dsm, non_blocking=True)
        g.replay()
        return out


def _modules_of(fn):
    """nn.Modules a forward callable reaches: itself / its bound object, and whatever its closure holds (directly or as `.model`)."""
    found, cand = [], [fn, getattr(fn, "__self__", None)]
    for cell in getattr(fn, "__closure__", None) or ():
        try:
            cand.append(cell.cell_contents)
        except ValueError:
            pass
    for c in cand:
        for o in (c, getattr(c, "model", None)):
            if isinstance(o, torch.nn.Module) and all(o is not f for f in found):
                found.append(o)
    return found


# ---------------------------------------------------------------------------------------------- evaluator-style
class SlidingWindowPredictor:
    """On-device `Evaluator._predict_image` (evaluator.py:113-176)."""

    def __init__(self, forward: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], n_classes: int,
                 window_size=(256, 256), batch_size: int = 16, group=None, accumulate=torch.float64, graph: bool = False):
        """accumulate: dtype of the merged-score raster. float64 (default) is the reference's numpy accumulator
        (evaluator.py:134): sums of window logits are exact, so the mask does not depend on the order the windows land in.
        float32 halves the raster (8192 x 8192 x 6 classes: 1.6 instead of 3.2 GB) at the price of order-dependent last bits."""
        self.forward, self.n_classes = (GraphedForward(forward) if graph else forward), n_classes      # graph: see GraphedForward
        self.window_size, self.batch_size, self.group = tuple(window_size), batch_size, group
        if accumulate not in (torch.float32, torch.float64):
            raise ValueError("accumulate: torch.float32 or torch.float64")
        self.accumulate = accumulate
        self._ebank = EvalBank()           # weight-derived operands kept between this predictor's forwards (bank.derived)

    @torch.no_grad()
    def predict(self, img: torch.Tensor, dsm: torch.Tensor, stride: int, return_scores: bool = False):
        """img [3,H,W] fp32 on the device (already scaled as in training), dsm [H,W] fp32 raw.
        Returns the uint8 class map [H,W] (identical on every rank) and, if asked, this rank's summed
        logits for the rows it owns (`(scores [C, rows, W], row_lo)`)."""
        C, (h, w) = self.n_classes, self.window_size
        _, H, W = img.shape
        if W % 4 or w % 4:
            raise ValueError("raster width and window width must be multiples of 4")
        rank, world = _dist_info(self.group)
        coords = [(x, y) for x, y, _, _ in sliding_window((H, W), stride, (h, w))]
        plan = plan_bands([c[0] for c in coords], [h] * len(coords), H, world)
        if rank >= len(plan):               # more ranks than windows
            me = dict(first=0, last=0, lo=0, hi=0, own_lo=0, own_hi=0)
        else:
            me = plan[rank]
        mm = ops.minmax(dsm.contiguous())                                   # whole-raster min / max (evaluator.py:100-111)
        d = ops.scale_shift(dsm.contiguous(), mm, 0)
        band = torch.zeros((C, max(1, me["hi"] - me["lo"]), W), dtype=self.accumulate, device=img.device)
        mine = coords[me["first"]:me["last"]]
        # ONE host -> device copy of the whole window list: a per-batch `torch.tensor(..., device=)` is a pageable copy that waits for
        # everything enqueued before it, i.e. it serialised the host's work of batch i + 1 behind the GPU's work of batch i
        c4_all = _coords4([(r, c, h, w) for r, c in mine], H, W, h, w, img.device)
        c2_all = (c4_all[:, :2] - torch.tensor([me["lo"], 0], dtype=torch.int32, device=img.device)).contiguous()
        for i0 in range(0, len(mine), self.batch_size):
            c4, c2 = c4_all[i0:i0 + self.batch_size], c2_all[i0:i0 + self.batch_size]
            rgb_t, dsm_t = ops.extract_tiles(img, d, c4, h, w)
            with self._ebank:
                logits = self.forward(rgb_t, dsm_t).float().contiguous()
            ops.tile_accumulate(band, logits, c2, atomic=True)              # pred[x:x+w, y:y+h] += out (evaluator.py:169-171)
        if world > 1 and rank < len(plan):
            merge_bands(band, plan, rank, _device_add_rows, self.group)
        mask = torch.zeros((H, W), dtype=torch.uint8, device=img.device)
        own = band[:, me["own_lo"] - me["lo"]:me["own_hi"] - me["lo"], :].contiguous()
        if own.shape[1] > 0:
            mask[me["own_lo"]:me["own_hi"]] = ops.argmax_chw(own)
        if world > 1:
            _share_owned_rows(plan, self.group, mask)                       # owned row ranges are disjoint: each travels once
        return (mask, (own, me["own_lo"])) if return_scores else mask


class Evaluator:
    """evaluator.py:22-291 with in-memory rasters (file I/O is host code outside the hot path):
    `_predict_image(net, img (H,W,3) float in [0,1], dsm (H,W), stride, batch_size, window_size)` returns
    the numpy class map like the reference; `evaluate(model_wrapper, samples, stride)` returns mIoU over
    the first 5 classes (utils2.py:168-216) from an on-device confusion matrix."""

    def __init__(self, n_classes: int = 6, window_size=(256, 256), batch_size: int = 10, graph: bool = True):
        self.n_classes, self.window_size, self.batch_size, self.graph = n_classes, tuple(window_size), batch_size, graph

    def _predict_image(self, net, img, dsm, stride, batch_size=None, window_size=None, device="cuda"):
        fwd = lambda a, b: net.forward(a, b, mode="Test") if hasattr(net, "forward") else net(a, b)      # noqa: E731
        p = SlidingWindowPredictor(fwd, self.n_classes, window_size or self.window_size, batch_size or self.batch_size)
        it = torch.as_tensor(np.ascontiguousarray(np.asarray(img, dtype=np.float32).transpose(2, 0, 1)), device=device)
        dt = torch.as_tensor(np.asarray(dsm, dtype=np.float32), device=device)
        return p.predict(it, dt, stride).cpu().numpy()

    def evaluate(self, model_wrapper, samples: Iterable[Tuple[np.ndarray, np.ndarray, np.ndarray]], stride: int = 64, device="cuda"):
        cm = torch.zeros((self.n_classes, self.n_classes), dtype=torch.int64, device=device)
        model_wrapper.set_eval_mode()
        # one predictor for the whole call: its operand store and its HIP graphs (one per batch shape) serve every sample. At the
        # reference's batch size of 10 a forward is host-bound when enqueued eagerly (2070 against 2810 windows/s replayed).
        p = SlidingWindowPredictor(lambda a, b: model_wrapper.forward(a, b, mode="Test"), self.n_classes, self.window_size, self.batch_size,
                                   graph=self.graph)
        for img, dsm, gt in samples:
            it = torch.as_tensor(np.ascontiguousarray(np.asarray(img, dtype=np.float32).transpose(2, 0, 1)), device=device)
            mask = p.predict(it, torch.as_tensor(np.asarray(dsm, dtype=np.float32), device=device), stride)
            cm += ops.confusion(mask.reshape(-1).contiguous(), torch.as_tensor(np.asarray(gt, dtype=np.int64), device=device).reshape(-1), self.n_classes)
        return metrics_from_confusion(cm.cpu().numpy())["mIoU"]


def metrics_from_confusion(cm: np.ndarray):
    """utils2.py:177-216 (OA, per-class F1, kappa, mIoU over the first 5 classes) from a confusion matrix."""
    cm = np.asarray(cm, dtype=np.float64)
    total = cm.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        oa = 100.0 * np.trace(cm) / total
        f1 = 2.0 * np.diag(cm) / (cm.sum(1) + cm.sum(0))
        pa = np.trace(cm) / total
        pe = np.sum(cm.sum(0) * cm.sum(1)) / (total * total)
        kappa = (pa - pe) / (1 - pe)
        iou = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
    return dict(cm=cm, OA=oa, F1=f1, kappa=kappa, mIoU=float(np.nanmean(iou[:5])))


# ---------------------------------------------------------------------------------------------- inference.py-style
class RasterInference:
    """Tile loop of `BuildingInference` (inference.py:498-568, 687-744) with an on-device overlap blend."""

    def __init__(self, forward: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], window_size=(256, 256),
                 overlap_ratio: float = 0.25, batch_size: int = 16, building_class: int = 1, threshold: float = 0.5, group=None,
                 graph: bool = False):
        self.forward, self.window_size = (GraphedForward(forward) if graph else forward), tuple(window_size)
        self.overlap = int(self.window_size[0] * overlap_ratio)                 # inference.py:139-141
        self.batch_size, self.cls, self.threshold, self.group = batch_size, building_class, threshold, group
        self._ebank = EvalBank()           # weight-derived operands kept between this predictor's forwards (bank.derived)

    @staticmethod
    def normalize_image(image: torch.Tensor, is_rgb: bool = True, nodata_value=None):
        """`_normalize_image` (inference.py:444-496) on the device: -> (fp32 image, bool nodata mask [H,W])."""
        return ops.normalize_raster(image, is_rgb, nodata_value)

    @staticmethod
    def _rgb(rgb):
        return ops.normalize_raster(rgb, True, None)[0] if rgb.dtype in (torch.uint8, torch.uint16) else rgb.float().contiguous()

    @torch.no_grad()
    def predict_tiles(self, rgb: torch.Tensor, dem: torch.Tensor, want_prob: bool = False):
        """The reference's own post step (`_tile_by_tile_vector_inference`, inference.py:713-757): per window, in the
        reference's window order, yields `((y, x, y_end, x_end), mask uint8 [wh, ww] numpy[, prob fp32 numpy])` with
        mask = softmax(logits)[building] >= threshold, cropped to the window - the input of `_create_tile_vectors`.
        Only the uint8 masks cross to the host (one copy per batch); ranks take contiguous slices of the window list."""
        h, w = self.window_size
        img = self._rgb(rgb)
        _, H, W = img.shape
        demc = dem.float().contiguous()
        rank, world = _dist_info(self.group)
        wins = get_sliding_windows(H, W, self.window_size, self.overlap)
        per = (len(wins) + world - 1) // world
        mine = wins[rank * per:(rank + 1) * per]
        c4_all = _coords4([(y, x, ye - y, xe - x) for y, x, ye, xe in mine], H, W, h, w, img.device)
        for i0 in range(0, len(mine), self.batch_size):
            chunk, c4 = mine[i0:i0 + self.batch_size], c4_all[i0:i0 + self.batch_size]
            rgb_t, dem_t = ops.extract_tiles(img, demc, c4, h, w)
            ops.tile_minmax_norm(dem_t)
            with self._ebank:
                logits = self.forward(rgb_t, dem_t).float().contiguous()
            prob, mask = ops.tile_prob_mask(logits, c4, self.cls, self.threshold, want_prob)
            mask_h = mask.cpu().numpy()
            prob_h = prob.cpu().numpy() if want_prob else None
            for i, (y, x, ye, xe) in enumerate(chunk):
                m = mask_h[i, :ye - y, :xe - x]
                yield ((y, x, ye, xe), m, prob_h[i, :ye - y, :xe - x]) if want_prob else ((y, x, ye, xe), m)

    @torch.no_grad()
    def predict(self, rgb: torch.Tensor, dem: torch.Tensor):
        """rgb [3,H,W] uint8 (or fp32 already normalised), dem [H,W] fp32 raw, both on the device.
        -> (prob [H,W] fp32: mean building probability over the covering tiles, mask [H,W] uint8)."""
        h, w = self.window_size
        img = self._rgb(rgb)
        _, H, W = img.shape
        rank, world = _dist_info(self.group)
        wins = get_sliding_windows(H, W, self.window_size, self.overlap)
        order = sorted(range(len(wins)), key=lambda i: (wins[i][0], wins[i][1]))
        wins = [wins[i] for i in order]
        plan = plan_bands([t[0] for t in wins], [t[2] - t[0] for t in wins], H, world) if wins else []
        me = plan[rank] if rank < len(plan) else dict(first=0, last=0, lo=0, hi=0, own_lo=0, own_hi=0)
        # [fixed-point prob sum ; tile count] as int64: integer adds commute, so the blend is bit-identical whatever the order
        # of the windows, the atomics and the ranks (ops.tile_prob_accumulate)
        band = torch.zeros((2, max(1, me["hi"] - me["lo"]), W), dtype=torch.int64, device=img.device)
        demc = dem.float().contiguous()
        mine = wins[me["first"]:me["last"]]
        c4_all = _coords4([(y, x, ye - y, xe - x) for y, x, ye, xe in mine], H, W, h, w, img.device)      # one copy for the whole list
        c4b_all = c4_all.clone()
        c4b_all[:, 0] -= me["lo"]
        for i0 in range(0, len(mine), self.batch_size):
            c4, c4b = c4_all[i0:i0 + self.batch_size], c4b_all[i0:i0 + self.batch_size]
            rgb_t, dem_t = ops.extract_tiles(img, demc, c4, h, w)              # reflect-pads short tiles
            ops.tile_minmax_norm(dem_t)                                        # per-tile DSM min-max
            with self._ebank:
                logits = self.forward(rgb_t, dem_t).float().contiguous()
            ops.tile_prob_accumulate(band[0], band[1], logits, c4b, self.cls)
        if world > 1 and rank < len(plan):
            merge_bands(band, plan, rank, _int_add_rows, self.group)
        prob = torch.zeros((H, W), dtype=torch.float32, device=img.device)
        mask = torch.zeros((H, W), dtype=torch.uint8, device=img.device)
        if me["own_hi"] > me["own_lo"]:
            sl = slice(me["own_lo"] - me["lo"], me["own_hi"] - me["lo"])
            p, m = ops.blend_finish(band[0, sl].contiguous(), band[1, sl].contiguous(), self.threshold)
            prob[me["own_lo"]:me["own_hi"]], mask[me["own_lo"]:me["own_hi"]] = p, m
        if world > 1:
            _share_owned_rows(plan, self.group, prob, mask)
        return prob, mask
