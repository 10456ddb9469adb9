#!/usr/bin/env python3
"""Time K5 (rba_swin_window_attn_f32) on the four Swin-B stage shapes of a 1024x2048 image.
(the round-1 RBA_K5_WAVES variant hook was removed with the losing variants)."""
import os
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops

busy = torch.randn(8192, 8192, device="cuda")
stages = [(256, 512, 4), (128, 256, 8), (64, 128, 16), (32, 64, 32)]
ws = 12
tot = 0.0
line = []
for (H, W, nH), reps, weight in zip(stages, (5, 5, 10, 10), (2, 2, 18, 2)):
    C = nH * 32
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(1, H * W, 3 * C, device="cuda", generator=g)
    qb = torch.randn(3 * C, device="cuda", generator=g) * 0.1
    bias = torch.randn(nH, 144, 144, device="cuda", generator=g) * 0.5
    frag = ops.swin_bias_fragments(bias, ws)
    for shift in (0, 6):
        ts = []
        for i in range(reps + 2):
            # ten launches queued behind a long kernel: the events bracket GPU time, not the host's launch latency (round 3)
            busy @ busy
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = ops.swin_window_attn(qkv, qb, bias, H, W, nH, ws, shift, bias_frag=frag)
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1) * 1e2)
        ts.sort()
        line.append(f"{H}x{W}/s{shift}: {ts[len(ts) // 2]:6.1f}us")
        tot += ts[len(ts) // 2] * weight / 2
print("  ".join(line) + f"  | per-image total {tot / 1e3:.2f} ms")
