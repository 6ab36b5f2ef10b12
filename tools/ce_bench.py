"""micro-timing of the loss kernels at the bench shape (16 x 6 x 256 x 256 logits)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops
lg = torch.randn(16, 6, 256, 256, device="cuda")
tg = torch.randint(0, 6, (16, 256, 256), device="cuda")
w = torch.ones(6, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("ce_fwd %.1f us" % t(lambda: ops.ce_loss_fwd(lg, tg, w, -100)))
