#!/usr/bin/env python3
"""A/B of the round-3 K6 kernel (split_linear_h3q.h: 128 x 64 sub-tiles, epilogue deferred into the next sub-tile's k loop) against the round-2
pipelined 128 x 128 kernel on the split-operand Linear shapes of Swin-B / Swin-L stages 3-4: bit-identity of the results and HIP-event time per
launch.  `python tools/k6_h3q_ab.py [swin_b|swin_l|c5] [reps]`"""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import _lib, ops  # noqa: E402


def shapes(which):
    if which == "swin_l":
        T3, C3, T4, C4 = 8192, 768, 2048, 1536
    elif which == "c5":
        T3, C3, T4, C4 = 3680, 512, 920, 1024
    else:
        T3, C3, T4, C4 = 8192, 512, 2048, 1024
    out = []
    for tag, T, C in (("s3", T3, C3), ("s4", T4, C4)):
        out += [(f"{tag} qkv", T, 3 * C, C, "f32"), (f"{tag} proj", T, C, C, "res"), (f"{tag} fc1", T, 4 * C, C, "gelu_split"),
                (f"{tag} fc2", T, C, 4 * C, "res")]
    return out


def run(mode, xs, lin, res):
    if mode == "f32":
        return ops.linear(xs, lin)
    if mode == "res":
        r = res.clone()
        return ops.linear(xs, lin, residual=r)
    return ops.linear(xs, lin, gelu=True, split_out=True).data


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return 1e3 * ts[len(ts) // 2]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    variant = ctypes.c_int.in_dll(_lib.load(), "rba_k6_variant")
    g = torch.Generator().manual_seed(0)
    tot = [0.0, 0.0]
    for name, M, N, K, mode in shapes(which):
        x = torch.randn(M, K, generator=g).cuda()
        lin = torch.nn.Linear(K, N).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        xs = ops.SplitActivations.pack(x)
        outs, us = [], []
        for v in (1, 2):
            variant.value = v
            outs.append(run(mode, xs, lin, res).clone())
            if mode == "res":
                r = res.clone()
                us.append(timeit(lambda: ops.linear(xs, lin, residual=r), reps))
            else:
                us.append(timeit(lambda: run(mode, xs, lin, res), reps))
        variant.value = 0
        same = torch.equal(outs[0], outs[1])
        ref = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
        err = None
        if mode != "gelu_split":
            want = ref + res.double() if mode == "res" else ref
            err = (outs[1].double() - want).abs().max().item()
        flops = 2.0 * M * N * K
        tot[0] += us[0]; tot[1] += us[1]
        print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d} {mode:10s} h3p {us[0]:7.1f} us  h3q {us[1]:7.1f} us  ({us[0] / us[1]:.2f}x)  "
              f"h3q {flops / us[1] / 1e6:6.1f} TF fp32-equiv = {3 * flops / us[1] / 1e6 / 2500 * 100:4.1f} % of f16 peak  bit-identical {same}"
              + (f"  max|err vs fp64| {err:.2e}" if err is not None else ""), flush=True)
    print(f"sum: h3p {tot[0]:.1f} us  h3q {tot[1]:.1f} us")
    if "--probe" in sys.argv:
        # ablations of the sub-tile kernel on fc1 (results wrong by construction): 1 no A loads, 2 no weight staging, 4 no deferred epilogue,
        # 8 no barrier, 16 no fragment reads
        M, N, K = (8192, 2048, 512) if which != "swin_l" else (8192, 3072, 768)
        x = torch.randn(M, K, generator=g).cuda()
        lin = torch.nn.Linear(K, N).cuda()
        xs = ops.SplitActivations.pack(x)
        for pr in (0, 1, 2, 4, 8, 16, 3, 5, 7, 15, 31):
            variant.value = 100 + pr
            print(f"fc1 h3q probe {pr:2d}: {timeit(lambda: run('gelu_split', xs, lin, None), reps):.1f} us", flush=True)
        variant.value = 0
    # stagger experiment on the round-2 kernel: the second workgroup of every CU starts late (ticks of the 100 MHz clock)
    stagger = ctypes.c_int.in_dll(_lib.load(), "rba_k6_stagger")
    variant.value = 1
    for name, M, N, K, mode in shapes(which):
        if ((M + 127) // 128) * ((N + 127) // 128) <= 256:
            continue
        x = torch.randn(M, K, generator=g).cuda()
        lin = torch.nn.Linear(K, N).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        xs = ops.SplitActivations.pack(x)
        row = []
        for d in (0, 200, 400, 600, 800, 1000, 1400):
            stagger.value = d
            row.append(f"{d}: {timeit(lambda: run(mode, xs, lin, res), reps):.1f}")
        stagger.value = 0
        print(f"{name:8s} {mode:10s} h3p with stagger (ticks: us)  " + "  ".join(row), flush=True)
    variant.value = 0


if __name__ == "__main__":
    main()
