"""Training side of the MFNet hot path: the train step of the reference's `Trainer.train_epoch`
(trainer.py:146-197; ignore-255 / skip-empty variant train_multiscale_main.py:113-171) as an
MI355X-native engine, plus drop-ins for `ModelWrapper` (model_wrapper.py:10-137) and `Trainer`
(trainer.py:26-465) that keep the reference's constructor / method surface.

What changes relative to the reference loop (the arithmetic of one step does not):
  * every tensor that receives a gradient lives in ONE flat fp32 parameter buffer with a matching flat
    gradient buffer and momentum buffer: `zero_grad` is one memset, `SGD.step` one fused kernel
    (mfnet_sgd_flat) instead of 333 (MMLoRA) / 357 (MMAdapter) small ones;
  * cross-entropy runs on the NCHW logits directly (mfnet_ce_loss_fwd/bwd: no transpose copy,
    utils2.py:115-127), and also yields the arg-max accuracy of the WHOLE batch on the device, so the
    reference's per-step `loss.item()` x3 + full-logit D2H (trainer.py:163-175) becomes one 16-byte read
    every `sync_every` steps;
  * data parallel: one process per GPU, the flat gradient buffer is cut into two buckets in backward
    order (decoder+FPN+fusion first, encoder PEFT second); each bucket is all-reduced (RCCL, SUM) as
    soon as its last gradient has been accumulated, overlapping the remaining encoder backward, and the
    1/world average is folded into the SGD kernel. SAM's prompt encoder / mask decoder never receive a
    gradient and are not part of any bucket. BatchNorm uses per-replica statistics.
"""
from __future__ import annotations

import os
import math
import time
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .bank import ParamBank, weights_changed

_ALIGN = 64          # elements: every tensor starts on a 256-byte boundary inside the flat buffers


# --------------------------------------------------------------------------------------------------
# loss (drop-in for utils2.CrossEntropy2d and the ignore_index variant)
# --------------------------------------------------------------------------------------------------
class _CE2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ignore_index):
        logits = logits.contiguous()
        acc, _ = ops.ce_loss_fwd(logits, target.contiguous(), weight, ignore_index)
        ctx.save_for_backward(logits, target, acc)
        ctx.weight, ctx.ignore = weight, ignore_index
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, g):
        logits, target, acc = ctx.saved_tensors
        d = ops.ce_loss_bwd(logits, target.contiguous(), acc, ctx.weight, ctx.ignore, 1.0)
        return d * g, None, None, None


def CrossEntropy2d(input, target, weight=None, size_average=True, ignore_index=-100):
    """utils2.py:115-127 (mean over pixels; `weight` = per-class weights) and, with ignore_index=255,
    train_multiscale_main.py:30-43. input (B,C,H,W) fp32 logits, target (B,H,W) int64."""
    if input.dim() != 4:
        raise ValueError("Expected 4 dimensions (got {})".format(input.dim()))
    if not size_average:
        raise NotImplementedError("size_average=False is never used by MFNet")
    return _CE2d.apply(input.float(), target, weight, int(ignore_index))


# --------------------------------------------------------------------------------------------------
# engine
# --------------------------------------------------------------------------------------------------
def trainable_tensors(model: nn.Module) -> List[Tuple[str, nn.Parameter]]:
    """Parameters that receive a gradient in MFNet's step, in bucket order: everything outside the
    encoder first (its backward runs first), encoder PEFT tensors second. `sam.prompt_encoder` /
    `sam.mask_decoder` are requires_grad but unreachable (unetformer_mmsam.py:533-563)."""
    head, enc = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad or n.startswith(("sam.prompt_encoder.", "sam.mask_decoder.")):
            continue
        (enc if ".image_encoder." in "." + n else head).append((n, p))
    return head + enc, len(head)


class TrainEngine:
    """One MFNet training step on one GPU (optionally one rank of a data-parallel group)."""

    def __init__(self, model: nn.Module, lr: float = 0.01, momentum: float = 0.9, weight_decay: float = 5e-4,
                 milestones: Sequence[int] = (25, 35, 45), gamma: float = 0.1, class_weight: Optional[torch.Tensor] = None,
                 ignore_index: int = -100, process_group=None, overlap: bool = True, use_graph: bool = False,
                 exchange_when_alone: bool = False, sync_bn: bool = False):
        self.model = model
        self.exchange_when_alone = bool(exchange_when_alone)      # world 1 still calls the collective (rehearsals)
        self.base_lr, self.momentum, self.weight_decay = float(lr), float(momentum), float(weight_decay)
        self.milestones, self.gamma = sorted(int(m) for m in milestones), float(gamma)
        self.epoch = 0
        self.ignore_index = int(ignore_index)
        self.class_weight = class_weight
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        # use_graph: zero_grad + forward + loss + backward are captured once per input shape into a HIP graph
        # and replayed (the step is ~2300 kernel launches: eager host enqueue time is 23 ms of a 29 ms step).
        # The gradient exchange and the SGD kernel (whose lr changes with the schedule) stay outside the graph.
        self.use_graph = bool(use_graph)
        # sync_bn: BatchNorm batch statistics over the whole data-parallel group (the strict 1-vs-N mode; default is
        # per-replica statistics like DDP). Its collectives sit between kernels of the step, so the step is enqueued eagerly.
        self.sync_bn = bool(sync_bn) and self.world > 1
        if self.sync_bn and self.use_graph:
            raise ValueError("sync_bn exchanges statistics inside the step: use_graph=False")
        self.overlap = overlap and not self.use_graph        # autograd hooks do not run when a graph is replayed
        # exchange_events: when a list, _finish_exchange appends an event pair (current stream) around the gradient collectives of
        # each step - bench.py's `allreduce_ms_per_step`; None costs nothing
        self.exchange_events = None
        self._graphs = {}
        self.keep_graph = False          # True: the captured hipGraph_t stays readable (graph_census.census(engine.captured_graph(...)))
        self._eager_warmup = 2
        named, n_head = trainable_tensors(model)
        if not named:
            raise ValueError("model has no trainable tensors")
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets, self.total = offs, off
        split = offs[n_head] if n_head < len(offs) else off
        self.bucket_bounds = [(0, split), (split, off)] if 0 < split < off else [(0, off)]
        self.bucket_of = [0 if (i < n_head or len(self.bucket_bounds) == 1) else 1 for i in range(len(self.params))]
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev) if momentum else None
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                if p.dtype != torch.float32:
                    raise TypeError("MFNet trains fp32 master parameters")
                v = self.flat_p[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        if self.world > 1:
            # identical replicas from the start (what DDP's constructor does): rank 0's parameters and BatchNorm buffers
            # replace whatever the other ranks initialised, so averaged gradients keep the replicas identical
            self.broadcast_state()
        self._pending = [0] * len(self.bucket_bounds)
        self._works = []
        self._launched = [False] * len(self.bucket_bounds)
        self._seen = set()
        self._bucket_size = [self.bucket_of.count(b) for b in range(len(self.bucket_bounds))]
        self._hooks = []
        # Gradients are added into flat_g by the modules' backward passes themselves (bank.deliver: one gather
        # launch per module instead of one autograd `grad += g` per tensor); the bucket counters are notified
        # from there. The autograd hook stays for any tensor whose gradient still arrives the ordinary way.
        self.bank = ParamBank()
        # overlapped exchange: buckets need their gradients at once. MFNET_EAGER_DELIVERY=1 is the A/B switch of the measurement.
        self.bank.defer = not (self.world > 1 and self.overlap) and os.environ.get("MFNET_EAGER_DELIVERY", "0") != "1"
        self.arena = ops.ZeroArena()
        for i, p in enumerate(self.params):
            notify = None
            if self.world > 1 and not self.use_graph:
                notify = self._make_hook(self.bucket_of[i])
                self._hooks.append(p.register_post_accumulate_grad_hook(notify))
            p._mfnet_sink = (p.grad, notify)
        self.steps = 0

    # ---- schedule (optim.lr_scheduler.MultiStepLR, trainer.py:118-122)
    @property
    def lr(self) -> float:
        return self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.epoch)

    def scheduler_step(self):
        self.epoch += 1

    # ---- data-parallel gradient exchange
    def _make_hook(self, b):
        def hook(_p):
            # a tensor is counted ONCE per step: a gradient handed over by bank.deliver notifies from there (its gather is
            # enqueued at that point), and autograd's accumulate hook still fires afterwards for the same tensor
            if id(_p) in self._seen:
                return
            self._seen.add(id(_p))
            self._pending[b] += 1
            if self._pending[b] == self._bucket_size[b] and self.overlap:
                self._launch(b)
        return hook

    def broadcast_state(self, src: int = 0):
        """Rank `src`'s trainable parameters and every floating / integer buffer (BatchNorm running statistics,
        num_batches_tracked) go to all ranks: called at construction and before a checkpoint is written (per-replica
        BatchNorm statistics differ between ranks; the saved model carries rank `src`'s)."""
        if self.world == 1:
            return
        root = dist.get_global_rank(self.pg, src) if self.pg is not None else src
        dist.broadcast(self.flat_p, src=root, group=self.pg)
        weights_changed()
        for b in self.model.buffers():
            if b.numel():
                dist.broadcast(b, src=root, group=self.pg)

    def _launch(self, b):
        lo, hi = self.bucket_bounds[b]
        self._launched[b] = True
        self._works.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _finish_exchange(self):
        if self.world == 1 and not (self.exchange_when_alone and dist.is_initialized()):
            return
        ev = None
        if self.exchange_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if not self.overlap and not self._works:
            # nothing to overlap with (graph replay / overlap off): ONE all-reduce over the whole flat gradient buffer -
            # every collective costs a launch and a stream hand-off of its own
            self._works.append(dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            # a bucket whose hooks did not all fire (a tensor without gradient this step) is reduced whole; exactly the
            # buckets that have not been launched yet, whatever order the others went in
            for b in range(len(self.bucket_bounds)):
                if not self._launched[b]:
                    self._launch(b)
        for w in self._works:
            w.wait()                      # stream-level wait: the compute stream waits for RCCL, the host does not
        if ev is not None:
            ev[1].record()
            self.exchange_events.append(ev)
        self._works.clear()
        self._pending = [0] * len(self.bucket_bounds)
        self._launched = [False] * len(self.bucket_bounds)
        self._seen.clear()

    # ---- one step
    def zero_grad(self):
        self.flat_g.zero_()

    def forward_backward(self, rgb, dsm, target, want_pred=False):
        """-> (acc [4] device tensor: sum w*nll, sum w, #correct, #valid ; pred uint8 | None)."""
        ops.SYNC_BN = None
        if self.sync_bn:
            # the group's batch size, exchanged once per step as an integer (ranks may hold ragged last batches): every BatchNorm
            # derives its group-wide row count from it on the host (ops._sync_sums)
            nb = torch.tensor([int(rgb.shape[0])], dtype=torch.int64, device=rgb.device)
            dist.all_reduce(nb, group=self.pg)
            ops.SYNC_BN = (self.pg, int(rgb.shape[0]), int(nb.item()))
        try:
            with self.bank, self.arena:   # bank: refreshes every operand derived from trainable weights, then serves them
                out = self.model(rgb, dsm, mode="Train")
                out = out.contiguous()
                acc, pred = ops.ce_loss_fwd(out, target, self.class_weight, self.ignore_index, want_pred=want_pred)
                d = ops.ce_loss_bwd(out, target, acc, self.class_weight, self.ignore_index, 1.0)
                out.backward(d)
        finally:
            ops.SYNC_BN = None
        return acc, pred

    def optimizer_step(self):
        self._finish_exchange()
        ops.sgd_flat(self.flat_p, self.flat_g, self.flat_m, self.lr, self.momentum, self.weight_decay, 1.0 / self.world)
        weights_changed()                 # the kernel rewrites the parameters without touching their version counters

    def step(self, rgb: torch.Tensor, dsm: torch.Tensor, target: torch.Tensor):
        """zero_grad -> forward -> CE -> backward (-> all-reduce) -> SGD. Returns the device tensor
        [sum w*nll, sum w, #correct, #valid] of THIS rank's batch; nothing is synchronised. In graph mode
        the returned tensor is overwritten by the next step of the same input shape."""
        if self.use_graph:
            acc = self._graph_forward_backward(rgb, dsm, target)
        else:
            self.zero_grad()
            acc, _ = self.forward_backward(rgb, dsm, target)
        self.optimizer_step()
        self.steps += 1
        return acc

    def _graph_forward_backward(self, rgb, dsm, target):
        key = (tuple(rgb.shape), tuple(dsm.shape), tuple(target.shape), self.model.training)
        ent = self._graphs.get(key)
        if ent is None:
            if self._eager_warmup > 0:           # builds the frozen-weight caches and kernel attributes outside any capture
                self._eager_warmup -= 1
                self.zero_grad()
                return self.forward_backward(rgb, dsm, target)[0]
            srgb, sdsm, stgt = rgb.clone(), dsm.clone(), target.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(keep_graph=True) if self.keep_graph else torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls its events while this thread captures; under the default
            # (global) mode that poll is an illegal call during capture and takes the process down
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.zero_grad()
                acc, _ = self.forward_backward(srgb, sdsm, stgt)
            ent = self._graphs[key] = (g, srgb, sdsm, stgt, acc)
        g, srgb, sdsm, stgt, acc = ent
        if rgb.data_ptr() != srgb.data_ptr():
            srgb.copy_(rgb, non_blocking=True)
        if dsm.data_ptr() != sdsm.data_ptr():
            sdsm.copy_(dsm, non_blocking=True)
        if target.data_ptr() != stgt.data_ptr():
            stgt.copy_(target, non_blocking=True)
        g.replay()
        return acc

    @property
    def grad_bytes(self) -> int:
        """bytes one rank puts into the gradient all-reduce per step (the flat fp32 gradient buffer)"""
        return int(self.flat_g.numel() * self.flat_g.element_size())

    def captured_graphs(self):
        """the torch.cuda.CUDAGraph objects captured so far, one per input shape (keep_graph=True makes them inspectable)."""
        return [ent[0] for ent in self._graphs.values()]

    @staticmethod
    def read_metrics(acc: torch.Tensor) -> Dict[str, float]:
        a = acc.tolist()                  # the only host sync
        return {"loss": a[0] / a[1] if a[1] else float("nan"), "accuracy": 100.0 * a[2] / a[3] if a[3] else float("nan")}

    def state_dict(self):
        """optimizer state for a true resume (the reference saves weights only, trainer.py:379-394)."""
        return {"momentum": None if self.flat_m is None else self.flat_m.clone(), "epoch": self.epoch, "steps": self.steps,
                "names": list(self.names)}

    def load_state_dict(self, sd):
        if sd.get("names") != self.names:
            raise ValueError("optimizer state belongs to a different parameter layout")
        if self.flat_m is not None and sd["momentum"] is not None:
            self.flat_m.copy_(sd["momentum"])
        self.epoch, self.steps = int(sd["epoch"]), int(sd["steps"])


# --------------------------------------------------------------------------------------------------
# reference API mirrors
# --------------------------------------------------------------------------------------------------
@dataclass
class TrainingConfig:
    """config.py:25-40."""
    base_lr: float = 0.01
    momentum: float = 0.9
    weight_decay: float = 0.0005
    scheduler_milestones: List[int] = None
    scheduler_gamma: float = 0.1
    batch_size: int = 10
    window_size: Tuple[int, int] = (256, 256)
    n_classes: int = 6
    cache: bool = True

    def __post_init__(self):
        if self.scheduler_milestones is None:
            self.scheduler_milestones = [25, 35, 45]


class ModelWrapper:
    """model_wrapper.py:10-137: builds `UNetFormer(num_classes).cuda()`, counts parameters by the
    'lora_' name test, saves / loads plain state dicts. Extra keyword arguments go to UNetFormer."""

    def __init__(self, num_classes: int = 6, **model_kwargs):
        from .unetformer_mmsam import UNetFormer
        self.num_classes = num_classes
        self.model = UNetFormer(num_classes=num_classes, **model_kwargs).cuda()

    def get_parameter_counts(self) -> Dict[str, int]:
        total = sum(p.nelement() for _, p in self.model.named_parameters())
        enc = lora = 0
        for n, p in self.model.image_encoder.named_parameters():
            if "lora_" not in n:
                enc += p.nelement()
            else:
                lora += p.nelement()
        return {"total": total, "image_encoder": enc, "lora": lora, "others": total - enc - lora}

    def print_parameter_summary(self) -> None:
        c = self.get_parameter_counts()
        print("=" * 50 + "\nMODEL PARAMETER SUMMARY\n" + "=" * 50)
        print(f"Total Parameters:     {c['total']:,}\nImage Encoder:        {c['image_encoder']:,}")
        print(f"LoRA Parameters:      {c['lora']:,}\nOther Parameters:     {c['others']:,}\n" + "=" * 50)

    def get_model(self) -> nn.Module:
        return self.model

    def load_weights(self, checkpoint_path: str, strict: bool = False) -> None:
        sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict):                      # inference.py:161-166 accepts wrapped formats too
            sd = sd.get("model_state_dict", sd.get("state_dict", sd))
        self.model.load_state_dict(sd, strict=strict)
        print(f"Successfully loaded weights from: {checkpoint_path}")

    def save_weights(self, save_path: str) -> None:
        torch.save(self.model.state_dict(), save_path)
        print(f"Model weights saved to: {save_path}")

    def set_train_mode(self) -> None:
        self.model.train()

    def set_eval_mode(self) -> None:
        self.model.eval()

    def forward(self, image_patches, dsm_patches, mode: str = "Train"):
        return self.model(image_patches, dsm_patches, mode=mode)


class Trainer:
    """trainer.py:26-465 with the data loader injected (dataset I/O is host code outside the hot path).

    `Trainer(model_wrapper, evaluator=None, train_loader=..., training=TrainingConfig(), epochs=50,
    save_epoch=1, ignore_index=-100, skip_empty=False)`; `.train_epoch(epoch) -> {'avg_loss',
    'training_time'}`, `.validate(epoch)`, `.save_checkpoint(epoch, miou)`, `.train(epochs, save_epoch)`.
    `skip_empty=True` reproduces MultiscaleTrainer's "skip batches without a building pixel" rule
    (train_multiscale_main.py:113-125), which costs one host sync per batch as in the reference.
    Under torch.distributed every rank runs the same loop on its shard of the batches; metrics are
    all-reduced when they are read, and only rank 0 prints / saves."""

    def __init__(self, model_wrapper: ModelWrapper, evaluator=None, *, train_loader: Iterable = None,
                 training: Optional[TrainingConfig] = None, epochs: int = 50, save_epoch: int = 1,
                 ignore_index: int = -100, skip_empty: bool = False, output_dir: str = "./results/",
                 model_name: str = "UNetformer", sync_every: int = 10, validate_fn=None, val_samples=None, test_ids=None,
                 val_stride: int = 32, use_graph: bool = True, sync_bn: bool = False):
        """sync_bn: BatchNorm batch statistics over the whole data-parallel group (TrainEngine; eager steps, so it turns use_graph off).
        use_graph: the engine replays zero_grad + forward + loss + backward from a HIP graph per batch shape (11.7 instead of
        19.0 ms per ViT-B MMLoRA step of 16 tiles: enqueued eagerly the ~640 launches of a step are host-bound). One graph per
        distinct batch shape (a shorter last batch captures its own); pass False for loaders whose shapes keep changing."""
        self.model_wrapper = model_wrapper
        self.model = model_wrapper.get_model()
        self.evaluator = evaluator
        self.training = training or TrainingConfig()
        self.train_loader = train_loader
        self.epochs, self.save_epoch = epochs, save_epoch
        self.best_miou = 0.82                                      # trainer.py:42
        self.skip_empty = skip_empty
        self.output_dir, self.model_name = output_dir, model_name
        self.sync_every = max(1, int(sync_every))
        self.validate_fn = validate_fn
        self.val_samples, self.test_ids, self.val_stride = val_samples, test_ids, val_stride
        if evaluator is not None and validate_fn is None and val_samples is None and test_ids is None:
            raise ValueError("Trainer: an evaluator needs its validation inputs - val_samples= or test_ids= (or give validate_fn=)")
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        dev = next(self.model.parameters()).device
        t = self.training
        # WEIGHTS = ones(N_CLASSES) (utils2.py:13); the ignore-255 variant uses no weights
        w = torch.ones(t.n_classes, device=dev) if ignore_index == -100 else None
        self.engine = TrainEngine(self.model, lr=t.base_lr, momentum=t.momentum, weight_decay=t.weight_decay,
                                  milestones=t.scheduler_milestones, gamma=t.scheduler_gamma, class_weight=w,
                                  ignore_index=ignore_index, use_graph=use_graph and not sync_bn, sync_bn=sync_bn)
        self.losses: List[float] = []
        self.recent_losses: List[float] = []
        self.recent_accuracies: List[float] = []
        self.loss_trend_window = 20

    def _reduce(self, acc):
        if dist.is_initialized() and dist.get_world_size() > 1:
            acc = acc.clone()
            dist.all_reduce(acc)
        return acc

    def train_epoch(self, epoch: int) -> Dict[str, float]:
        self.model.train()
        start = time.time()
        dev = self.engine.flat_p.device
        pend = torch.zeros(4, device=dev)
        epoch_acc = torch.zeros(4, device=dev)
        nb = 0
        multi = dist.is_initialized() and dist.get_world_size() > 1
        for batch_idx, (data, dsm, target) in enumerate(self.train_loader):
            if self.skip_empty:
                # train_multiscale_main.py:113-125. Under data parallelism the decision is COLLECTIVE: a step is skipped
                # only when every rank's batch is empty, so all ranks issue the same gradient collectives. Labels still in host
                # memory (a plain DataLoader) are inspected there: on the device the answer costs a host sync per step.
                valid = target[target != 255]
                flag = 1 if (valid.numel() == 0 or bool(torch.all(valid == 0))) else 0
                if multi:
                    empty = torch.tensor([flag], device=dev, dtype=torch.int32)
                    dist.all_reduce(empty, op=dist.ReduceOp.MIN)
                    flag = int(empty.item())
                if flag:
                    continue
            data = data.to(dev, non_blocking=True)
            dsm = dsm.to(dev, non_blocking=True)
            target = target.to(dev, non_blocking=True)
            acc = self.engine.step(data, dsm, target)
            pend += acc
            epoch_acc += acc
            nb += 1
            # metric syncs follow batch_idx (equal on every rank given equal loader lengths), never a per-rank count
            if (batch_idx + 1) % self.sync_every == 0:
                m = TrainEngine.read_metrics(self._reduce(pend))
                pend.zero_()
                if m["loss"] == m["loss"]:
                    self.losses.append(m["loss"])
                    self.recent_losses = (self.recent_losses + [m["loss"]])[-self.loss_trend_window:]
                    self.recent_accuracies = (self.recent_accuracies + [m["accuracy"]])[-self.loss_trend_window:]
                    if self.rank == 0:
                        print(f"\rEpoch {epoch:2d}/{self.epochs} {batch_idx:4d} | Loss: {m['loss']:7.4f} | "
                              f"Acc: {m['accuracy']:5.1f}% | LR: {self.engine.lr:.2e}", end="")
                    self._check_early_overfitting_signs()
        m = TrainEngine.read_metrics(self._reduce(epoch_acc))
        if self.rank == 0:
            print()
        return {"avg_loss": m["loss"], "training_time": time.time() - start}

    def _check_early_overfitting_signs(self):
        """trainer.py:295-319: stop when the recent mean batch accuracy exceeds 99 % (rank-consistent:
        the accuracies it looks at are already all-reduced)."""
        if len(self.recent_accuracies) >= 5 and sum(self.recent_accuracies[-5:]) / 5 > 99.0:
            raise KeyboardInterrupt("auto-stop: recent training accuracy above 99 %")

    def validate(self, epoch: int) -> float:
        """trainer.py:345-377. `validate_fn(model_wrapper)` wins if given; otherwise the evaluator is called the way its
        interface asks: the drop-in `ssrs_amd.inference.Evaluator` with the in-memory `val_samples` (and `val_stride`), an
        evaluator with the reference's file-based signature with `test_ids=` / `stride=` (evaluator.py:178-230)."""
        if self.validate_fn is None and self.evaluator is None:
            print("No evaluator provided, skipping validation")
            return 0.0
        self.model.eval()
        try:
            if self.validate_fn is not None:
                miou = self.validate_fn(self.model_wrapper)
            elif self.val_samples is not None:
                miou = self.evaluator.evaluate(self.model_wrapper, self.val_samples, stride=self.val_stride)
            elif self.test_ids is not None:
                miou = self.evaluator.evaluate(self.model_wrapper, test_ids=self.test_ids, stride=self.val_stride)
            else:
                raise ValueError("Trainer.validate: pass val_samples= (in-memory rasters for ssrs_amd.inference.Evaluator), "
                                 "test_ids= (an evaluator with the reference's signature) or validate_fn= to Trainer")
        finally:
            self.model.train()
        return float(miou)

    def save_checkpoint(self, epoch: int, miou: float) -> None:
        """trainer.py:379-394. Holds NO collective: under data parallelism `train()` has already made the replicas
        identical (`broadcast_state`) and the mIoU rank 0's, so every rank takes the same branch."""
        if miou > self.best_miou:
            self.best_miou = miou
            if self.rank == 0:
                import os
                os.makedirs(self.output_dir, exist_ok=True)
                self.model_wrapper.save_weights(f"{self.output_dir}{self.model_name}_epoch{epoch}_{miou:.4f}")
        elif self.rank == 0:
            print(f"   No improvement (best: {self.best_miou:.4f})")

    def _rank0_value(self, value: float) -> float:
        """rank 0's float on every rank (one 8-byte broadcast; nothing to do alone)."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.engine.flat_p.device)
        dist.broadcast(t, src=0)
        return float(t.item())

    def train(self, epochs: Optional[int] = None, save_epoch: Optional[int] = None) -> None:
        epochs = self.epochs if epochs is None else epochs
        save_epoch = self.save_epoch if save_epoch is None else save_epoch
        for epoch in range(1, epochs + 1):
            t0 = time.time()
            metrics = self.train_epoch(epoch)
            self.engine.scheduler_step()
            if self.rank == 0:
                print(f"EPOCH {epoch:2d}/{epochs}  train {metrics['training_time']:.2f}s  total {time.time() - t0:.2f}s  "
                      f"avg loss {metrics['avg_loss']:.4f}  lr {self.engine.lr:.2e}")
            if epoch % save_epoch == 0:
                # Data parallel: per-replica BatchNorm running statistics differ between ranks, so rank 0's state
                # goes to every rank BEFORE the validation - the mIoU then describes the model that is saved, whichever rank
                # evaluated which band of windows. The collectives are unconditional (same sequence on every rank), and the
                # best-mIoU comparison runs on rank 0's value so no rank can take the other branch.
                self.engine.broadcast_state()
                miou = self._rank0_value(self.validate(epoch))
                self.save_checkpoint(epoch, miou)
