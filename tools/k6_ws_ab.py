#!/usr/bin/env python3
"""K6 weights-stationary EXPERIMENT (csrc/tune/split_linear_ws.hip, librba_tune.so) against the product's pipelined kernel on the same split-image operands; warm
(one buffer) and cold (a ring of input buffers larger than the Infinity Cache) operands; the stagger between the first tiles of a SIMD's waves.
Result: profiles/r04_k6_ws.txt (not adopted).   python tools/k6_ws_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import _lib, ops

import _tune
tl = _tune.load()
stg = ctypes.c_int.in_dll(tl, "rba_k6_ws_stagger")
tl.rba_ws_launch.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
tl.rba_ws_launch.restype = ctypes.c_int
SHAPES = (("s3 fc1", 8192, 2048, 512, "gelu_split"), ("s3 proj", 8192, 512, 512, "res"), ("s2 fc1", 32768, 1024, 256, "gelu_split"), ("s2 proj", 32768, 256, 256, "res"),
          ("s1 proj", 131072, 128, 128, "res"), ("plain", 8192, 1024, 512, "plain"))
for cold in (False, True):
    for name, M, N, K, kind in SHAPES:
        torch.manual_seed(0)
        nbuf = max(2, int(600e6 // (M * K * 4))) if cold else 1
        xs = [ops.SplitActivations.pack(torch.randn(M, K, device="cuda")) for _ in range(nbuf)]
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda")
        planes = ops.split_weight(w, mode="f16x3")

        out = torch.empty(M, N, device="cuda")
        sout = ops.SplitActivations.empty((M, N), out.device)

        def call_ws(i):
            x = xs[i % nbuf]
            st = torch.cuda.current_stream().cuda_stream
            if kind == "plain":
                rc = tl.rba_ws_launch(0, 0, x.data.data_ptr(), planes.data_ptr(), b.data_ptr(), 0, out.data_ptr(), M, N, K, st)
            elif kind == "res":
                rc = tl.rba_ws_launch(1, 0, x.data.data_ptr(), planes.data_ptr(), b.data_ptr(), r.data_ptr(), r.data_ptr(), M, N, K, st)
            else:
                rc = tl.rba_ws_launch(2, 1, x.data.data_ptr(), planes.data_ptr(), b.data_ptr(), 0, sout.data.data_ptr(), M, N, K, st)
            _lib.check(rc, "ws")

        def call(i):
            x = xs[i % nbuf]
            if kind == "plain":
                return ops.split_linear(x, planes, b, out_features=N)
            if kind == "res":
                return ops.split_linear(x, planes, b, out_features=N, residual=r)
            return ops.split_linear(x, planes, b, gelu=True, out_features=N, split_out=True)
        cfgs = ((0, 0), (2, 0), (2, 150), (2, 300), (2, 600))
        ts = {c: [] for c in cfgs}
        reps = max(nbuf, 5)
        for rnd in range(7):
            for c in cfgs:
                stg.value = c[1]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    (call_ws if c[0] else call)(i)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ts[c].append(e0.elapsed_time(e1) * 1000.0 / reps)
        stg.value = 300
        med = {c: sorted(t)[len(t) // 2] for c, t in ts.items()}
        flops = 6.0 * M * N * K
        print(f"{'cold' if cold else 'warm'} {name:8s} M={M} N={N} K={K} {kind:10s} pipelined {med[(0, 0)]:6.1f} us | weights-stationary, stagger 0 / 150 / 300 / 600: "
              f"{med[(2, 0)]:6.1f} {med[(2, 150)]:6.1f} {med[(2, 300)]:6.1f} {med[(2, 600)]:6.1f} us  (best {flops / min(med[c] for c in cfgs[1:]) / 1e6:.0f} TFLOP/s f16)", flush=True)
