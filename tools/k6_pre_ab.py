#!/usr/bin/env python3
"""What the pre-split A operand (the LayerNorm kernel's fragment image, "PRE") is worth to the stage-3 GEMMs that follow a LayerNorm: the same launch fed
fp32 rows (split in the k loop) against the split image.  Prices the LayerNorm-folded form (round 5, docs/kernels/K6.md).  Usage: tools/k6_pre_ab.py [streams_hint]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops

hint = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops.set_concurrent_streams(hint)
busy = torch.randn(8192, 8192, device="cuda")


def timed(fn, reps=7, inner=20):
    ts = []
    for i in range(reps + 2):
        busy @ busy
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


torch.manual_seed(0)
for (M, K, N, gelu) in ((8192, 512, 1536, False), (8192, 512, 2048, True), (8192, 768, 2304, False), (8192, 768, 3072, True)):
    lin = torch.nn.Linear(K, N).cuda()
    x = torch.randn(M, K, device="cuda")
    g, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    with torch.no_grad():
        xs = ops.add_layer_norm(x, g, b, 1e-5, frag=True)[1]
        xf = xs.unpack()
        so = gelu
        t_rows = timed(lambda: ops.linear(xf, lin, gelu=gelu, split_out=so))
        t_pre = timed(lambda: ops.linear(xs, lin, gelu=gelu, split_out=so))
        t_ln = timed(lambda: ops.add_layer_norm(x, g, b, 1e-5, frag=True))
    print(f"hint={hint} M={M} K={K} N={N} gelu={gelu}: fp32 rows {t_rows:6.1f} us   split image {t_pre:6.1f} us   (the LayerNorm that writes it: {t_ln:5.1f} us)", flush=True)
