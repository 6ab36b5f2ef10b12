"""Run-to-run reproducibility of the train step (tiny MMLoRA model): 60 evaluations of forward + backward at identical
weights agree to fp32 noise (no race, no uninitialised read); a one-ulp change of 8 fp32 weights leaves the logits bit-identical
yet moves the gradients by ~4e-3 relative L2, growing stage by stage up the backward chain: every bf16 rounding of an
intermediate turns a perturbation d into about sqrt(d * ulp_bf16) RMS. Run on the GPU box: python tools/grad_noise.py"""
import sys, io, warnings, argparse
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, numpy as np
from helpers import rel_err
from ssrs_amd import detfill, ops
from ssrs_amd.trainer import TrainEngine
from ssrs_amd.unetformer_mmsam import UNetFormer
def tiny():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return UNetFormer(num_classes=6, encoder="vit_tiny_test", args=argparse.Namespace(mod="sam_lora", mid_dim=None, image_size=1024, multimask_output=1))
fill = detfill.fill_state_dict(tiny().state_dict(), "ckpt/")
rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("ckpt", 2, 256, 6)]
m = tiny(); m.load_state_dict(fill); m = m.cuda().train(); m.decoder.segmentation_head[1].p = 0.0
e = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=False)
outs = {}
def grads():
    e.zero_grad()
    with e.bank, e.arena:
        out = e.model(rgb, dsm, mode="Train").contiguous()
        acc, pred = ops.ce_loss_fwd(out, tgt, e.class_weight, e.ignore_index, want_pred=False)
        d = ops.ce_loss_bwd(out, tgt, acc, e.class_weight, e.ignore_index, 1.0)
        out.backward(d)
    torch.cuda.synchronize()
    return e.flat_g.clone(), out.detach().clone()
for _ in range(2): grads(); e.optimizer_step()
g0, o0 = grads()
nbad = 0
for it in range(60):
    g, o = grads()
    dg, do = float(rel_err(g, g0)), float(rel_err(o, o0))
    if dg > 1e-5 or do > 1e-6:
        nbad += 1
        per = []
        for n, p, off in zip(e.names, e.params, e.offsets):
            a, b = g[off:off + p.numel()], g0[off:off + p.numel()]
            per.append((float((a - b).abs().max() / (b.abs().max() + 1e-30)), n))
        print(it, "grad", dg, "logits", do, sorted(per, reverse=True)[:3])
print("deviating evaluations:", nbad, "of 60")
# sensitivity: one-ulp changes of a few weights
import itertools
for trial in range(6):
    saved = e.flat_p.clone()
    o = e.offsets[e.names.index("decoder.b2.attn.proj.0.weight")]
    idx = torch.randint(0, 64 * 64, (8,), device="cuda") + o
    e.flat_p[idx] = torch.nextafter(e.flat_p[idx], torch.full_like(e.flat_p[idx], 1e9))
    g, out = grads()
    per = []
    for n, p, off in zip(e.names, e.params, e.offsets):
        a, b = g[off:off + p.numel()], g0[off:off + p.numel()]
        per.append((float((a - b).abs().max() / (b.abs().max() + 1e-30)), n))
    print("ulp trial", trial, "grad", float(rel_err(g, g0)), "logits", float(rel_err(out, o0)), sorted(per, reverse=True)[:2])
    e.flat_p.copy_(saved)
# control: version bump without any change
o = e.offsets[e.names.index("decoder.b2.attn.proj.0.weight")]
idx = torch.arange(8, device="cuda") + o
e.flat_p[idx] = e.flat_p[idx].clone()
g, out = grads()
print("no-op write: grad", float(rel_err(g, g0)), "logits", float(rel_err(out, o0)))
# which entries carry the deviation of a one-ulp trial: relative L2 over the whole gradient and over the encoder only
saved = e.flat_p.clone()
e.flat_p[idx] = torch.nextafter(e.flat_p[idx], torch.full_like(e.flat_p[idx], 1e9))
g, out = grads()
print("one-ulp: rel L2 of whole gradient", float((g - g0).norm() / g0.norm()), "logits", float(rel_err(out, o0)))
for n, p, off in zip(e.names, e.params, e.offsets):
    a, b = g[off:off + p.numel()], g0[off:off + p.numel()]
    r = float((a - b).norm() / (b.norm() + 1e-30))
    if r > 1e-3 and p.numel() > 64:
        print(f"   L2 {r:.2e} |g| {float(b.norm()):.2e} {n}")
