"""Where the cycles of the persistent bf16x6 Linear (v5) go: per-workgroup cycle counters of MFMA wave 0 (barrier wait / compute /
epilogue) and loader wave 0 (vmcnt wait / barrier wait / DMA issue).  python tools/gemm_v5_timing.py M N K cfg"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import _tune
from _tune import ops

M, N, K, cfg = (int(v) for v in sys.argv[1:5])
fn = _tune.load().rba_split_linear_v5_timing
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * K ** -0.5
b = torch.randn(N, device="cuda")
planes = ops.split_weight(w)
out = torch.empty(M, N, device="cuda")
dbg = torch.zeros(1024 * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    rc = fn(x.data_ptr(), planes.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, cfg, dbg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
torch.cuda.synchronize()
d = dbg.view(-1, 8).cpu().double()
d = d[d[:, 3] > 0]
names = ["mfma: barrier wait", "mfma: compute", "mfma: epilogue", "mfma: total", "loader: vmcnt wait", "loader: barrier wait", "loader: issue", "loader: total"]
print(f"M={M} N={N} K={K} cfg={cfg}: {d.shape[0]} workgroups; cycle-counter ticks per workgroup (mean / min / max)")
for i, n in enumerate(names):
    print(f"  {n:22s} {d[:, i].mean():10.0f} {d[:, i].min():10.0f} {d[:, i].max():10.0f}   {100 * d[:, i].mean() / d[:, 3 if i < 4 else 7].mean():5.1f} %")
ref = (x[:256].double() @ w.double().T + b.double())
print("max err rows 0..255:", (out[:256].double() - ref).abs().max().item())
