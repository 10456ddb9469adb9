#!/usr/bin/env python3
"""K6 f16x3 kernel (csrc/split_linear_h3.h) against the bf16x6 product kernel on every token-Linear shape of a backbone:
time (HIP events, interleaved rounds) and error against an fp64 reference, beside hipBLASLt's fp32 GEMM.

  python tools/gemm_h3_sweep.py [swin_b|swin_l|c5] [cfgs: 4,2,1 = tile width / 32; 14/24/44 ablations of fc1]"""
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _tune
from rba_amd import _lib, ops

lib = _tune.load()
fn = lib.rba_split_linear_h3_tune
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
fn.restype = ctypes.c_int


def shapes(which):
    if which == "c5":
        E, H, W = 128, 720, 1280
    else:
        E, H, W = (128 if which == "swin_b" else 192), 1024, 2048
    out = []
    for s in range(4):
        C = E << s
        M = (H >> (2 + s)) * (W >> (2 + s))
        out += [(f"s{s + 1} qkv", M, 3 * C, C, 0), (f"s{s + 1} proj", M, C, C, 0), (f"s{s + 1} fc1", M, 4 * C, C, 1), (f"s{s + 1} fc2", M, C, 4 * C, 0)]
    return out


def h3(x, planes, bias, act, cfg, N):
    K = x.shape[-1]
    M = x.numel() // K
    out = torch.empty(M, N, device=x.device)
    rc = fn(x.data_ptr(), planes.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, act, cfg, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, f"h3 cfg {cfg}")
    return out


which = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
cfgs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 2]
tot = {"bf16x6": 0.0, "blas": 0.0, "best": 0.0, "product": 0.0}
only = sys.argv[3] if len(sys.argv) > 3 else None
for name, M, N, K, act in shapes(which):
    if only and name != only.replace("_", " "):
        continue
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda") * torch.exp(torch.randn(M, K, device="cuda"))          # heavy-tailed, like GELU outputs
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    p6 = ops.split_weight(w, mode="bf16x6")
    p3 = ops.split_weight(w, mode="f16x3")
    runs = {"bf16x6": lambda: ops.split_linear(x, p6, b, gelu=bool(act), out_features=N), "product": lambda: ops.split_linear(x, p3, b, gelu=bool(act), out_features=N),
            "blas": lambda: (torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b)) if act else torch.nn.functional.linear(x, w, b))}
    for c in cfgs:
        runs[f"h3:{c}"] = (lambda c=c: h3(x, p3, b, act, c, N))
    times = {k: [] for k in runs}
    outs = {}
    for rnd in range(int(os.environ.get('ROUNDS', '7'))):
        for k, f in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = f(); e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) * 1e3)
            outs[k] = y
    rows = slice(0, min(M, 2048))
    ref = x[rows].double() @ w.double().T + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    sc = ref.abs().mean()
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    err = {k: float(((outs[k][rows].double() - ref).pow(2).mean().sqrt() / sc)) for k in outs}
    mx = {k: float(((outs[k][rows].double() - ref).abs().max() / sc)) for k in outs}
    best = min((k for k in med if k.startswith("h3:") and int(k[3:]) % 1000 < 10), key=lambda k: med[k], default="product")
    tot["bf16x6"] += med["bf16x6"]; tot["blas"] += med["blas"]; tot["best"] += med[best]; tot["product"] += med["product"]
    tf = 2.0 * M * N * K / med[best] / 1e6
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d} act={act} " + "  ".join(f"{k}: {med[k]:6.1f}" for k in med) +
          f" | best {best} {tf:6.1f} TF fp32-equiv ({3 * tf / 2500 * 100:.1f}% of f16 peak) | rms err " +
          " ".join(f"{k} {err[k]:.1e}" for k in ("blas", "bf16x6", best)) + " | max " + " ".join(f"{k} {mx[k]:.1e}" for k in ("blas", "bf16x6", best)), flush=True)
print("sum: " + "  ".join(f"{k} {v:.0f} us" for k, v in tot.items()))
