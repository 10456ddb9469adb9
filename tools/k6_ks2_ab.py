#!/usr/bin/env python3
"""K6 single-resident launches (exactly one 128 x 128 tile per CU: stage-3 proj / fc2 of Swin-B), round 4: the pipelined kernel with one wave per
SIMD (OCC = 2 build, what the product launches; cfg 5204) against the 8-wave forms -- K split over two wave sets with two weight rings (KS = 2, cfg 6104)
and the 256 x 128 row-split form (RS = 2, cfg 7104: half the CUs).  Timing-only tune-library launches on split-image operands.
python tools/k6_ks2_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import _lib, ops
import _tune

fn = _tune.load().rba_split_linear_h3_tune
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
for name, M, N, K in (("s3 proj", 8192, 512, 512), ("s3 fc2", 8192, 512, 2048), ("s3 qkv", 8192, 1536, 512), ("L s3 proj", 8192, 768, 768), ("L s3 fc2", 8192, 768, 3072)):
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda")
    xs = ops.SplitActivations.pack(x).data                      # a real split image (the data a launch sees decides its power draw and clock)
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    p3 = ops.split_weight(w, mode="f16x3")
    out = torch.empty(M, N, device="cuda")
    cfgs = (5204, 5104, 6104, 7104)
    ts = {c: [] for c in cfgs}
    for rnd in range(9):
        for c in cfgs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.check(fn(xs.data_ptr(), p3.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, c, torch.cuda.current_stream().cuda_stream), f"cfg {c}")
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts[c].append(e0.elapsed_time(e1) * 200.0)
    med = {c: sorted(v)[len(v) // 2] for c, v in ts.items()}
    print(f"{name:10s} M={M} N={N} K={K}  tiles {((M + 127) // 128) * ((N + 127) // 128):5d}   128x128 OCC=2 build {med[5204]:6.1f} us   OCC=1 build {med[5104]:6.1f}   KS=2 (8 waves, K split) {med[6104]:6.1f}"
          f"   RS=2 (256x128) {med[7104]:6.1f}", flush=True)
