#!/usr/bin/env python3
"""K5 at Swin-B stage 3 (16 heads): time per workgroup as a function of the workgroup count -- does the partial third round of the 1 056-workgroup launch
(66 windows x 16 heads on 512 resident slots) cost a whole round?   python tools/k5_tail_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops

busy = torch.randn(8192, 8192, device="cuda")
nH, ws = 16, 12
C = nH * 32
g = torch.Generator(device="cuda").manual_seed(0)
bias = torch.randn(nH, 144, 144, device="cuda", generator=g) * 0.5
frag = ops.swin_bias_fragments(bias, ws)
qb = torch.randn(3 * C, device="cuda", generator=g) * 0.1
for (H, W) in ((48, 96), (60, 120), (72, 96), (96, 96), (64, 128), (72, 132), (96, 120), (96, 144), (96, 192)):
    qkv = torch.randn(1, H * W, 3 * C, device="cuda", generator=g)
    nw = -(-H // ws) * -(-W // ws)
    for shift in (0, 6):
        ts = []
        for i in range(7):
            busy @ busy
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.swin_window_attn(qkv, qb, bias, H, W, nH, ws, shift, bias_frag=frag)
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1) * 1e2)
        ts.sort()
        t = ts[len(ts) // 2]
        print(f"{H:3d}x{W:3d} shift {shift}: {nw:3d} windows x {nH} heads = {nw * nH:5d} workgroups = {nw * nH / 512:.2f} rounds of 512: {t:6.1f} us, {t / (nw * nH) * 512:5.1f} us per 512 workgroups", flush=True)
