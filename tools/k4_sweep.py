#!/usr/bin/env python3
"""Time K4 (rba_mask_logits_f32) on the model shapes (the round-1 variant hook was removed with the losing variants)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops
for (h, w) in ((256, 512), (184, 320)):
    g = torch.Generator(device="cuda").manual_seed(0)
    e = torch.randn(1, 100, 256, device="cuda", generator=g)
    f = torch.randn(1, 256, h, w, device="cuda", generator=g)
    ref = torch.einsum("bqc,bchw->bqhw", e.double(), f.double())
    ts = []
    for i in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = ops.mask_logits(e, f); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    fl = 2 * 100 * 256 * h * w
    print(f"{h}x{w}: {ts[len(ts)//2]:7.1f} us  {fl / ts[len(ts)//2] / 1e6:6.1f} TFLOP/s  max|d| {(out.double() - ref).abs().max().item():.2e}")
