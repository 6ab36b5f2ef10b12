import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssrs_amd import ops
torch.manual_seed(0)
for (B, S) in ((16, 8), (16, 16), (16, 32)):
    heads = 8
    qkv = torch.randn(B * S * S, 3 * 64, device="cuda").to(torch.bfloat16)
    table = torch.randn(225, heads, device="cuda") * 0.1
    d_o = torch.randn(B * S * S, 64, device="cuda")
    for fn, name in ((lambda: ops.gla_attn_fwd(qkv, table, B, S, S, heads, 8 ** -0.5), "fwd"), (lambda: ops.gla_attn_bwd(qkv, table, d_o, B, S, S, heads, 8 ** -0.5), "bwd")):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"S={S} {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
