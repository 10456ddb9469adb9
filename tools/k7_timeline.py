#!/usr/bin/env python3
"""Per-wave timeline of K7 (timing build, csrc/tune/k7_timing.hip): where a window's workgroup spends its shader cycles.
  python tools/k7_timeline.py [H W shift]      (default: Swin-B stage 1 of a 1024x2048 image: 256 512 0)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import _tune
from rba_amd import _lib, ops

lib = _tune.load()
fn = lib.rba_k7_timing
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 6 + [ctypes.c_float] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
fn.restype = ctypes.c_int
H, W, shift = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 512, 0)
C, nH = 128, 4
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
x0 = r(1, H * W, C)
g1, b1, g2, b2 = r(C, sc=0.2) + 1, r(C, sc=0.2), r(C, sc=0.2) + 1, r(C, sc=0.2)
qw, qb, pw, pb = r(3 * C, C, sc=C ** -0.5), r(3 * C, sc=0.1), r(C, C, sc=C ** -0.5), r(C, sc=0.1)
bias = r(nH, 144, 144, sc=0.5)
frag = ops.swin_bias_fragments(bias, 12)
img = ops.swin_attn_block_weights(qw, pw)
Hp, Wp = (H + 11) // 12 * 12, (W + 11) // 12 * 12
nwg = (Hp // 12) * (Wp // 12)
y2 = torch.empty_like(x0)
for rep in range(3):
    x = x0.clone()
    dbg = torch.zeros(nwg * 9 * 32, dtype=torch.int64, device="cuda")
    rc = fn(x.data_ptr(), y2.data_ptr(), g1.data_ptr(), b1.data_ptr(), 1e-5, img.data_ptr(), qb.data_ptr(), frag.data_ptr(), pb.data_ptr(), g2.data_ptr(),
            b2.data_ptr(), 1e-5, 1, H, W, C, shift, dbg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "k7 timing")
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwg, 9, 32).astype(np.float64)
life = d[:, :, 31].max(axis=1) - d[:, :, 0].min(axis=1)
print(f"== K7 {H}x{W} shift {shift}: {nwg} workgroups (one window each); workgroup lifetime median {np.median(life) / 1e3:.2f} kticks of s_memtime "
      f"(p10 {np.percentile(life, 10) / 1e3:.2f}, p90 {np.percentile(life, 90) / 1e3:.2f})")


def show(name, a, b):
    v = (d[:, :, b] - d[:, :, a]).ravel()
    print(f"   {name:34s} median {np.median(v):8.0f} cyc   p10 {np.percentile(v, 10):8.0f}   p90 {np.percentile(v, 90):8.0f}")


show("x load + norm1 + split", 0, 1)
for h in range(nH):
    prev = 1 if h == 0 else 6 + 5 * (h - 1)
    show(f"head {h}: wait chunk + barrier B", prev, 2 + 5 * h)
    show(f"head {h}: q, k, v GEMM + LDS write", 2 + 5 * h, 3 + 5 * h)
    show(f"head {h}: barrier A", 3 + 5 * h, 4 + 5 * h)
    show(f"head {h}: attention", 4 + 5 * h, 5 + 5 * h)
    show(f"head {h}: proj", 5 + 5 * h, 6 + 5 * h)
show("epilogue (+ norm2)", 6 + 5 * (nH - 1), 31)
