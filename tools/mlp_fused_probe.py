#!/usr/bin/env python3
"""Ablations of the fused Swin MLP kernel (csrc/mlp_fused_h3.h PROBE bits: 1 no GELU, 2 no fc1 MFMAs, 4 no fc2 MFMAs, 8 no staging / barrier).
  python tools/mlp_fused_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _tune
from rba_amd import _lib, ops

lib = _tune.load()
fn = lib.rba_mlp_fused_probe
fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
fn.restype = ctypes.c_int
M, C, hidden = 131072, 128, 512
fc1, fc2 = torch.nn.Linear(C, hidden).cuda(), torch.nn.Linear(hidden, C).cuda()
x, r = torch.randn(M, C, device="cuda"), torch.randn(M, C, device="cuda")
p1, p2 = ops.split_weight(fc1.weight.detach(), mode="f16x3"), ops.split_weight(fc2.weight.detach(), mode="f16x3")
out = torch.empty_like(r)
for probe, what in ((0, "product"), (1, "no GELU"), (2, "no fc1 MFMAs"), (4, "no fc2 MFMAs"), (6, "no MFMAs"), (7, "no MFMAs, no GELU"), (8, "no staging / barrier"),
                    (9, "no staging, no GELU"), (14, "only GELU + split + swaps"), (15, "loop skeleton")):
    def run():
        _lib.check(fn(x.data_ptr(), p1.data_ptr(), fc1.bias.data_ptr(), p2.data_ptr(), fc2.bias.data_ptr(), r.data_ptr(), out.data_ptr(), M, hidden, probe,
                      torch.cuda.current_stream().cuda_stream), "probe")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"probe {probe:2d} ({what}): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
