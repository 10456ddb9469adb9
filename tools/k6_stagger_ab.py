#!/usr/bin/env python3
"""K6 short-K launches of Swin stages 1-2 (many rounds of 128 x 128 tiles, the epilogue as long as the k loop): does a late start of every CU's second
workgroup (rba_k6_stagger, ticks of the 100 MHz clock) de-phase the two workgroups' epilogues for the whole launch?  Product entry points on split-image
operands, cold operands (a ring of input buffers larger than the Infinity Cache).   python tools/k6_stagger_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import _lib, ops

stg = ctypes.c_int.in_dll(_lib.load(), "rba_k6_stagger")
SHAPES = (("s1 qkv", 131072, 384, 128, "plain"), ("s1 proj", 131072, 128, 128, "res"), ("s2 qkv", 32768, 768, 256, "plain"), ("s2 proj", 32768, 256, 256, "res"),
          ("s2 fc1", 32768, 1024, 256, "gelu_split"), ("s2 fc2", 32768, 256, 1024, "res"), ("s3 qkv", 8192, 1536, 512, "plain"))
for name, M, N, K, kind in SHAPES:
    torch.manual_seed(0)
    nbuf = max(2, int(600e6 // (M * K * 4)))
    xs = [ops.SplitActivations.pack(torch.randn(M, K, device="cuda")) for _ in range(nbuf)]
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    planes = ops.split_weight(w, mode="f16x3")

    def call(i):
        x = xs[i % nbuf]
        if kind == "plain":
            return ops.split_linear(x, planes, b, out_features=N)
        if kind == "res":
            return ops.split_linear(x, planes, b, out_features=N, residual=r)
        return ops.split_linear(x, planes, b, gelu=True, out_features=N, split_out=True)
    vals = (0, 200, 400, 700, 1000)
    ts = {v: [] for v in vals}
    for rnd in range(7):
        for v in vals:
            stg.value = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nbuf):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts[v].append(e0.elapsed_time(e1) * 1000.0 / nbuf)
    stg.value = 0
    med = {v: sorted(t)[len(t) // 2] for v, t in ts.items()}
    print(f"{name:8s} M={M} N={N} K={K} {kind:10s} " + "  ".join(f"stagger {v:4d}: {med[v]:6.1f} us" for v in vals), flush=True)
