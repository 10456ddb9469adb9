#!/usr/bin/env python3
"""The fused Swin MLP (C = 128) on ONE accumulator per output -- EXPERIMENT, csrc/tune/mlp_fused_h1.hip in librba_tune.so -- against the product's two-accumulator
kernel: error against fp64 on three data recipes, and time.  Result: profiles/r04_mlp_one_accumulator.txt (not adopted).   python tools/mlp_one_acc_ab.py"""
import ctypes
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import _lib, ops
import _tune

tl = _tune.load()
tl.rba_split_weight_f16x2_scaled.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
tl.rba_swin_mlp_fused_f16x3s_f32.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
for f in (tl.rba_split_weight_f16x2_scaled, tl.rba_swin_mlp_fused_f16x3s_f32):
    f.restype = ctypes.c_int


def scaled_planes(w):
    amax = float(w.abs().max())
    e = 13 - math.floor(math.log2(amax))
    N, K = w.shape
    packed = torch.empty(((N + 127) // 128, K // 16, 2, 128, 2, 8), dtype=torch.float16, device=w.device)
    _lib.check(tl.rba_split_weight_f16x2_scaled(w.data_ptr(), packed.data_ptr(), N, K, 2.0 ** e, torch.cuda.current_stream().cuda_stream), "split scaled")
    return packed, 2.0 ** -e


def one_acc(x, fc1, fc2, r, planes):
    (p1, i1), (p2, i2) = planes
    M, C = x.shape
    _lib.check(tl.rba_swin_mlp_fused_f16x3s_f32(x.data_ptr(), p1.data_ptr(), fc1.bias.data_ptr(), p2.data_ptr(), fc2.bias.data_ptr(), r.data_ptr(), r.data_ptr(), M, C,
                                                fc1.weight.shape[0], i1, i2, torch.cuda.current_stream().cuda_stream), "mlp one acc")
    return r


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


C = 128
for M, hidden, recipe in ((131072, 512, "plain"), (32768, 512, "heavy"), (32768, 256, "tiny")):
    g = torch.Generator().manual_seed(M + hidden)
    fc1, fc2 = torch.nn.Linear(C, hidden).cuda(), torch.nn.Linear(hidden, C).cuda()
    x, r = torch.randn(M, C, generator=g) * 2, torch.randn(M, C, generator=g)
    with torch.no_grad():
        if recipe == "heavy":
            x = x * torch.exp(torch.randn(1, C, generator=g) * 2.0)
            x[:, 5] *= 100.0
            x[:, 77] *= 1e-4
            fc1.weight.mul_((torch.exp(torch.randn(hidden, 1, generator=g) * 1.5) * 0.05).cuda())
            fc2.weight.mul_(torch.exp(torch.randn(C, 1, generator=g) * 1.5).cuda())
        if recipe == "tiny":
            x = x * 1e-3
        x, r = x.cuda(), r.cuda()
        ref = r.double() + F.linear(F.gelu(F.linear(x.double(), fc1.weight.double(), fc1.bias.double())), fc2.weight.double(), fc2.bias.double())
        planes = (scaled_planes(fc1.weight.detach().contiguous()), scaled_planes(fc2.weight.detach().contiguous()))
        one = one_acc(x, fc1, fc2, r.clone(), planes)
        two = ops.mlp_fused(x, fc1, fc2, r.clone())
        e = lambda y: (float((y.double() - ref).abs().max()), float((y.double() - ref).pow(2).mean().sqrt()))
        (m1, r1), (m2, r2) = e(one), e(two)
        t1 = t(lambda: one_acc(x, fc1, fc2, r, planes))
        t2 = t(lambda: ops.mlp_fused(x, fc1, fc2, r))
    print(f"{recipe:6s} M {M} hidden {hidden} |ref| max {float(ref.abs().max()):.3e}: one accumulator max {m1:.3e} rms {r1:.3e}, {t1:6.1f} us | two accumulators max {m2:.3e} rms {r2:.3e}, {t2:6.1f} us", flush=True)
