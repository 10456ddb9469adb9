#!/usr/bin/env python3
"""The decoder's 100-query Linears (rba_skinny_linear_f32): round 1-2 decomposition vs the per-row-tile one.  python tools/skinny_ab.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import _lib, ops
var = ctypes.c_int.in_dll(_lib.load(), "rba_skinny_variant")
for name, N, K in (("q / out proj", 256, 256), ("class head", 20, 256), ("FFN linear1", 2048, 256), ("FFN linear2", 256, 2048)):
    x, w, b = torch.randn(100, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5, torch.randn(N, device="cuda")
    res = {}
    for v in (1, 2):
        var.value = v
        for _ in range(3):
            ops.skinny_linear(x, w, b)
        evs = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.skinny_linear(x, w, b); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(c) for a, c in evs)
        res[v] = ts[len(ts) // 2] * 1e3
    var.value = 0
    print(f"{name:12s} 100 x {K} -> {N}: round 1-2 {res[1]:6.1f} us   per-row-tile {res[2]:6.1f} us (HIP events, incl. launch)")
