"""How far can ANY fp32 evaluation of the train step be from the float64 one? Runs the CPU oracle (oracle/mfnet_ref.py, pinned by
the reference's goldens at 1e-9 in float64) on the tiny-model train step in float32 and compares every gradient with the
reference's float64 record, exactly as tests/test_fp32_gpu.py measures the HIP fp32 path. Test infrastructure (imports oracle/);
`python tools/diag_fp32_grads.py [sam_lora|sam_adpt]`. Result (both mods): logits 5e-6, loss 1e-7 - and the gradients median
2e-3, worst 5e-2 (167 of 173 tensors beyond 1e-4): the ReLU6 / ReLU / max-pool switches and the 2-tile training-mode BatchNorm make
the gradient map discontinuous at fp32 resolution, for torch's own CPU kernels as for any other fp32 arithmetic."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from helpers import golden, oracle_train_step_grads, summary_err  # noqa: E402

if __name__ == "__main__":
    mod = sys.argv[1] if len(sys.argv) > 1 else "sam_lora"
    tag = f"model_tiny_{mod}"
    g = golden(tag)
    l32, o32, g32 = oracle_train_step_grads(tag, mod, torch.float32)
    print(f"loss fp64 {float(g['loss0']):.9f} fp32 {l32:.9f}; logits rel err "
          f"{float(np.abs(o32[:, :, ::4, ::4].double().numpy() - g['logits_s4']).max() / np.abs(g['logits_s4']).max()):.2e}")
    floor = 1e-5 * float(np.median([g["g/" + n][0] for n in g32]))
    rows = sorted(((max(summary_err(v, g["g/" + n], floor)), n, v.numel()) for n, v in g32.items()), reverse=True)
    print("worst ten (max of norm / leading-slice error, numel):")
    for e, n, k in rows[:10]:
        print(f"  {n}: {e:.2e}  ({k})")
    errs = np.array([r[0] for r in rows])
    print(f"{len(rows)} gradients: median {np.median(errs):.2e}, 90 % {np.quantile(errs, .9):.2e}, max {errs.max():.2e}; above 1e-4: {(errs > 1e-4).sum()}")
