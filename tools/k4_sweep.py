#!/usr/bin/env python3
"""Time K4 (rba_mask_logits_f32 = exact-fp32 MFMA, rba_mask_logits_f16x3_f32 = three f16 products per fp32 product) on the model shapes:
the dense heads of C2 (256 x 512 columns) and C5 (180 x 320) and the sparse intermediate heads (4 h w gathered columns per level)."""
import ctypes
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import _lib, ops
lib = _lib.load()
busy = torch.randn(8192, 8192, device="cuda")
knob = ctypes.c_int.in_dll(lib, "rba_k4_variant")
pf = ctypes.c_int.in_dll(lib, "rba_k4_waves")
for N in (256 * 512, 180 * 320, 4 * 32 * 64, 4 * 90 * 160, 4 * 45 * 80, 4 * 23 * 40):
    g = torch.Generator(device="cuda").manual_seed(0)
    e = torch.randn(1, 100, 256, device="cuda", generator=g)
    f = torch.randn(1, 256, N, device="cuda", generator=g)
    ref = torch.einsum("bqc,bcn->bqn", e.double(), f.double())
    line = [f"N={N:7d}"]
    for mode, var in (("fp32", 0), ("f16x3", 0), ("f16x3", 1), ("f16x3", 2), ("f16x3", 11), ("f16x3", 12)):
        knob.value = var % 10          # 1x / 2x: 4 waves per workgroup instead of 8
        pf.value = 4 if var > 10 else 8
        ts = []
        for i in range(6):
            # ten launches queued behind a long kernel, so that the events bracket GPU time and not the host's launch latency
            busy @ busy
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = ops.mask_logits(e, f, mode=mode)
            e1.record(); torch.cuda.synchronize()
            if i >= 2: ts.append(e0.elapsed_time(e1) * 1e2)
        ts.sort()
        line.append(f"{mode}{'/ct' + str(var) if var else ''} {ts[len(ts)//2]:6.1f} us (max|d| {(out.double() - ref).abs().max().item():.1e})")
    knob.value = 0
    pf.value = 8
    nbytes = 4 * (256 * N + 100 * N + 100 * 256)
    print("  ".join(line) + f"  | HBM floor {nbytes / 8e6:.1f} us")
