"""Node census of the captured train step + where the step's copies / fills come from.

  python tools/graph_census.py [tiny|vitb] [out.json]

(1) captures the step with keep_graph=True and prints ssrs_amd.graph_census.census(): node types and the edges of every node
    that is not a kernel; (2) runs one EAGER step under the torch profiler with stacks and lists, per python call site inside
    ssrs_amd/, the device-to-device copies and fills it enqueues (those become memcpy / memset nodes or blit kernels)."""
import argparse
import json
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import detfill                                          # noqa: E402
from ssrs_amd.graph_census import census                              # noqa: E402
from ssrs_amd.trainer import TrainEngine                              # noqa: E402
from ssrs_amd.unetformer_mmsam import UNetFormer                      # noqa: E402


def build(kind, mod):
    args = argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder="vit_tiny_test" if kind == "tiny" else "vit_b", args=args)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "census/"))
    return m.cuda().train()


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    mod = os.environ.get("MFNET_CENSUS_MOD", "sam_lora")
    B = 2 if kind == "tiny" else 16
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("census", B, 256, 6)]
    m = build(kind, mod)
    eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=True)
    eng.keep_graph = True
    for _ in range(4):
        eng.step(rgb, dsm, tgt)
    torch.cuda.synchronize()
    c = census(eng.captured_graphs()[0])
    res = {"model": kind, "mod": mod, "batch": B, "census": c}
    print(json.dumps({k: v for k, v in c.items() if k != "non_kernel"}))
    for n in c["non_kernel"][:40]:
        print("  non-kernel node", n)

    # eager step under the profiler: who enqueues copies / fills
    m2 = build(kind, mod)
    e2 = TrainEngine(m2, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=False)
    for _ in range(2):
        e2.step(rgb, dsm, tgt)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        e2.zero_grad()
        e2.forward_backward(rgb, dsm, tgt)
        torch.cuda.synchronize()
    sites = {}
    for ev in prof.events():
        nm = ev.name
        if nm in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat", "aten::stack",
                  "aten::index_add_", "aten::bernoulli_", "aten::div", "aten::mul", "aten::add", "aten::_to_copy"):
            st = [f for f in (ev.stack or []) if "ssrs_amd" in f]
            key = (nm, st[0] if st else "(outside ssrs_amd)")
            sites[key] = sites.get(key, 0) + 1
    rows = sorted(sites.items(), key=lambda kv: -kv[1])
    res["eager_aten_sites"] = [{"op": k[0], "site": k[1], "calls": v} for k, v in rows]
    for r in res["eager_aten_sites"][:60]:
        print(f"  {r['calls']:4d} x {r['op']:16s} {r['site']}")
    kern = {}
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA and ("copy" in ev.name.lower() or "fill" in ev.name.lower() or "Memcpy" in ev.name or "Memset" in ev.name):
            kern[ev.name[:90]] = kern.get(ev.name[:90], 0) + 1
    res["eager_copy_fill_kernels"] = kern
    print(json.dumps(kern, indent=1))
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
