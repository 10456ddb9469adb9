#!/usr/bin/env python3
"""Per-wave timeline of K5's f16x3 kernel (timing build, csrc/tune/k5_timing.hip): phase durations, workgroup lifetime, residency.
  python tools/k5_timeline.py [H W nH]      (default: Swin-B stage 3 of a 1024x2048 image: 64 128 16)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import _tune
from rba_amd import _lib, ops

lib = _tune.load()
fn = lib.rba_k5_timing
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_void_p]
fn.restype = ctypes.c_int
H, W, nH = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 128, 16)
C = nH * 32
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(1, H * W, 3 * C, device="cuda", generator=g)
qb = torch.randn(3 * C, device="cuda", generator=g) * 0.1
bias = torch.randn(nH, 144, 144, device="cuda", generator=g) * 0.5
frag = ops.swin_bias_fragments(bias, 12)
rows = (H * W + 31) // 32 * 32
out = torch.empty(rows * C, device="cuda")
Hp, Wp = (H + 11) // 12 * 12, (W + 11) // 12 * 12
nwg = (Hp // 12) * (Wp // 12) * nH
names = ["issue loads", "loads arrive", "split+LDS write", "barrier", "Q split+bias arrive", "QK^T", "softmax", "PV", "store"]
big = torch.randn(64 << 20, device="cuda")                 # 256 MB: evicts qkv from L2 / Infinity Cache for the cold leg
for shift, wpe in ((0, 5), (6, 5), (0, 6)):
    for cold in (0, 1):
        dbg = torch.zeros(nwg * 9 * 12, dtype=torch.int64, device="cuda")
        for _ in range(3):
            if cold:
                big.add_(1.0)
            rc = fn(qkv.data_ptr(), qb.data_ptr(), frag.data_ptr(), out.data_ptr(), 1, H, W, nH, shift, 1, wpe, dbg.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "k5 timing")
            torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(nwg, 9, 12)
        t = d[:, :, :10].astype(np.float64) / 100.0         # us
        t -= t[:, :, 0].min()
        life = t[:, :, 9].max(axis=1) - t[:, :, 0].min(axis=1)
        print(f"== {H}x{W} nH {nH} shift {shift} wpe {wpe} ({'96' if wpe == 5 else '80'} VGPRs) {'cold' if cold else 'warm'}: {nwg} workgroups, span {t[:, :, 9].max():.1f} us; workgroup lifetime median "
              f"{np.median(life):.2f} us (p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f})")
        ph = t[:, :, 1:10] - t[:, :, 0:9]
        for i, n in enumerate(names):
            v = ph[:, :, i].ravel()
            print(f"   {n:22s} median {np.median(v):5.2f} us  p10 {np.percentile(v, 10):5.2f}  p90 {np.percentile(v, 90):5.2f}")
        first = t[:, :, 0].min(axis=1)
        end = t[:, :, 9].max(axis=1)
        for ts in np.linspace(0.5, end.max() - 0.5, 10):
            alive = int(((first <= ts) & (end > ts)).sum())
            ing = int(((t[:, :, 0] <= ts) & (t[:, :, 4] > ts)).sum())
            print(f"   t={ts:5.1f} us: {alive:4d} workgroups resident; {ing:5d} waves in the gather phase, "
                  f"{int(((t[:, :, 4] <= ts) & (t[:, :, 9] > ts)).sum()):5d} in the compute phase")
        hw = d[:, :, 10]
        simd = (hw >> 4) & 3
        cu = ((hw >> 32) & 0xf) * 10000 + ((hw >> 13) & 0x7) * 1000 + ((hw >> 12) & 1) * 100 + ((hw >> 8) & 0xf)
        print(f"   distinct CUs {len(np.unique(cu[:, 0]))}; waves per SIMD of a workgroup (mean over workgroups): "
              f"{[round(float((simd == s).sum(axis=1).mean()), 2) for s in range(4)]}")
