#!/usr/bin/env python3
"""Per-workgroup timeline of the f16x3 kernel (timing build): residency, prologue / loop / epilogue durations.
  python tools/gemm_h3_timing.py M N K [probe]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import _tune
from rba_amd import _lib, ops

lib = _tune.load()
fn = lib.rba_split_linear_h3_timing
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
fn.restype = ctypes.c_int
M, N, K = (int(v) for v in sys.argv[1:4])
probe = int(sys.argv[4]) if len(sys.argv) > 4 else 0
x = torch.randn(M, K, device="cuda")
if probe in (1300, 1400, 1500, 1600, 1700):          # launch forms that read a split image
    x = ops.SplitActivations.pack(x).data
w = torch.randn(N, K, device="cuda") * K ** -0.5
b = torch.randn(N, device="cuda")
p3 = ops.split_weight(w, mode="f16x3")
out = torch.empty(M, N, device="cuda")
nwg = ((M + (255 if probe in (1600, 1700) else 127)) // (256 if probe in (1600, 1700) else 128)) * ((N + 127) // 128)
dbg = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda")
for _ in range(3):
    rc = fn(x.data_ptr(), p3.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, probe, dbg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "timing")
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwg, 6)
t = d[:, :4].astype(np.float64) / 100.0            # us
t0 = t[:, 0].min()
t -= t0
print(f"{nwg} workgroups; kernel span {t[:, 3].max():.1f} us")
print(f"prologue {np.median(t[:, 1] - t[:, 0]):.2f} us  loop {np.median(t[:, 2] - t[:, 1]):.2f} us  epilogue {np.median(t[:, 3] - t[:, 2]):.2f} us (medians);"
      f" loop p10/p90 {np.percentile(t[:, 2] - t[:, 1], 10):.2f}/{np.percentile(t[:, 2] - t[:, 1], 90):.2f}")
if probe >= 1000 and probe != 1001:
    lp = np.median(t[:, 2] - t[:, 1])
    print(f"shader clock in the loop: {np.median(d[:, 4]) / lp / 1e3:.2f} GHz; {np.median(d[:, 4]) / (K // 32):.0f} clk per 32-wide block (MFMA alone: 768)")
# residency: number of workgroups alive at sample times
for ts in np.linspace(1, t[:, 3].max() - 1, 12):
    alive = int(((t[:, 0] <= ts) & (t[:, 3] > ts)).sum())
    inloop = int(((t[:, 1] <= ts) & (t[:, 2] > ts)).sum())
    print(f"  t={ts:6.1f} us: {alive:4d} workgroups resident, {inloop:4d} in the k loop")
hw = d[:, 5]
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
key = d[:, 4] * 1000 + se * 100 + sh * 50 + cu
print("distinct (xcc, se, sh, cu) ids:", len(np.unique(key)), "; start-time histogram (us):", np.histogram(t[:, 0], bins=8)[0].tolist())
