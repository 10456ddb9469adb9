#!/usr/bin/env python3
"""K6, round 4: the 256 x 128 / eight-wave / shared-weight-ring form (RS = 2 of split_linear_h3p_kernel, tune library) against the product's
128 x 128 form, on split-image operands (timing-only launches of the tune library: cfg 5204 / 7104 = fp32 rows out, the qkv launch; 5214 / 7114 =
GELU + split image out, the fc1 launch), and -- for correctness -- on fp32 rows against fp64 (cfg 7004 vs the product).  Interleaved rounds,
HIP events, median.      python tools/k6_rs2_ab.py [swin_b|swin_l]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import _lib, ops
import _tune

import ctypes
fn = _tune.load().rba_split_linear_h3_tune
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
which = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
E = 128 if which == "swin_b" else 192
shapes = []
for s in (2, 3):
    C, M = E << s, (1024 >> (2 + s)) * (2048 >> (2 + s))
    shapes += [(f"s{s + 1} qkv", M, 3 * C, C, 0), (f"s{s + 1} proj", M, C, C, 0), (f"s{s + 1} fc1", M, 4 * C, C, 1), (f"s{s + 1} fc2", M, C, 4 * C, 0)]


def run(x, planes, bias, out, M, N, K, act, cfg):
    _lib.check(fn(x.data_ptr(), planes.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, act, cfg, torch.cuda.current_stream().cuda_stream), f"cfg {cfg}")


for name, M, N, K, act in shapes:
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    p3 = ops.split_weight(w, mode="f16x3")
    out = torch.empty(M, N, device="cuda")
    cfgs = (5214, 7114) if act else (5204, 7104)
    ts = {c: [] for c in cfgs}
    for rnd in range(9):
        for c in cfgs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(x, p3, b, out, M, N, K, act, c)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts[c].append(e0.elapsed_time(e1) * 200.0)
    med = {c: sorted(v)[len(v) // 2] for c, v in ts.items()}
    # correctness of the RS = 2 form on fp32 rows
    ref = x[:1024].double() @ w.double().T + b.double()
    ref = torch.nn.functional.gelu(ref) if act else ref
    run(x, p3, b, out, M, N, K, act, 7004)
    e_rs = float((out[:1024].double() - ref).abs().max())
    prod = ops.split_linear(x, p3, b, gelu=bool(act), out_features=N)
    e_pr = float((prod[:1024].double() - ref).abs().max())
    same = torch.equal(out, prod)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d} {'gelu_split' if act else 'f32       '} tiles {tiles:5d}  128x128 {med[cfgs[0]]:6.1f} us   256x128 RS=2 {med[cfgs[1]]:6.1f} us"
          f"  ({med[cfgs[0]] / med[cfgs[1]]:.2f}x)   fp32-rows form: bit-identical to the product {same}, max|err vs fp64| {e_rs:.2e} (product {e_pr:.2e})", flush=True)
