#!/usr/bin/env python3
"""Swin-B stage-1 MLP (C = 128, hidden 512): the one-kernel form against fc1 (split output) + fc2 (residual epilogue).  python tools/mlp_fused_ab.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rba_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
C, hidden = 128, 512
fc1, fc2 = torch.nn.Linear(C, hidden).cuda(), torch.nn.Linear(hidden, C).cuda()
x, r = torch.randn(M, C, device="cuda"), torch.randn(M, C, device="cuda")


def t(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    a = t(lambda: ops.linear(ops.linear(x, fc1, gelu=True, split_out=True), fc2, residual=r))
    b = t(lambda: ops.mlp_fused(x, fc1, fc2, r))
    want = ops.linear(ops.linear(x, fc1, gelu=True, split_out=True), fc2, residual=r.clone())
    got = ops.mlp_fused(x, fc1, fc2, r.clone())
fl = 2.0 * M * C * hidden * 2
print(f"M {M}: fc1 + fc2 {a:.1f} us   fused {b:.1f} us ({fl / b / 1e6:.0f} TFLOP/s fp32-equivalent)   equal {torch.equal(want, got)}")
