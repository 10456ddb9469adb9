import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops
x = torch.randn(131072, 256, device="cuda"); w = torch.randn(256, 256, device="cuda") / 16; b = torch.randn(256, device="cuda")
for mode in ("bf16x6", "f16x3"):
    pl = ops.split_weight(w, mode=mode)
    for _ in range(3): ops.split_linear_nchw_out(x, pl, b, 131072)
    evs = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.split_linear_nchw_out(x, pl, b, 131072); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) for a, c in evs)
    print(mode, round(ts[len(ts) // 2] * 1e3, 1), "us")
