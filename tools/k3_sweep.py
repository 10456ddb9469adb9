#!/usr/bin/env python3
"""Time K3 (rba_masked_xattn_f32): v1 (one workgroup per (query, head)) vs v2 (split-key matrix-pipe path)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops
for S in (100, 920, 2048, 3680, 14720):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(1, n, 8, 32, device="cuda", generator=g) for n in (100, S, S))
    ml = torch.randn(1, 100, S, device="cuda", generator=g) * 3
    res = []
    for split in (False, True):
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = ops.masked_xattn(q, k, v, ml, split_keys=split); e1.record(); torch.cuda.synchronize()
            if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); res.append((ts[len(ts) // 2], out))
    print(f"S={S:6d}: v1 {res[0][0]:7.1f} us   v2 {res[1][0]:7.1f} us   max|v1-v2| {(res[0][1] - res[1][1]).abs().max().item():.2e}")
