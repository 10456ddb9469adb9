#!/usr/bin/env python3
"""add_layer_norm: fp32 rows vs split-image output (and the fused residual add), per Swin stage shape.  python tools/layer_norm_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rba_amd import ops


def t(f, n=100):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, C in ((8192, 512), (2048, 1024), (8192, 768), (32768, 384), (2048, 1536), (32768, 256), (131072, 128)):
    x = torch.randn(rows, C, device="cuda")
    w, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    a = t(lambda: ops.add_layer_norm(x, w, b))
    f = t(lambda: ops.add_layer_norm(x, w, b, frag=True))
    y0 = ops.add_layer_norm(x, w, b)[1]
    y1 = ops.add_layer_norm(x, w, b, frag=True)[1]
    ok = torch.equal(y1.unpack(), ops.SplitActivations.pack(y0).unpack())
    print(f"rows {rows:6d} C {C:5d}: fp32 rows {a:6.1f} us   split image {f:6.1f} us   ({rows * C * 8 / f / 1e6:.2f} TB/s)  equal {ok}")
