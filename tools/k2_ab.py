#!/usr/bin/env python3
"""K2 (multi-scale deformable attention core) at BASELINE C5's encoder shape (19 320 queries, 3 levels of a 736 x 1280 image, 8 heads x 32) and at
C2's (2 048 queries, 1 level): the generic kernel (one thread per 4 channels, rolled tap loop) vs the round-3 kernel (32 consecutive queries of one
head per workgroup, compile-time L / P, 16 taps in flight) vs the fused form (sampling locations + softmax inside the kernel).  HIP-event medians.
Sampling geometry as in the network: reference points = pixel centres, offsets = ring bias + noise.   python tools/k2_ab.py [reps]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import _lib, ops  # noqa: E402
from rba_amd.seeded_weights import deform_ring_bias  # noqa: E402


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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    variant = ctypes.c_int.in_dll(_lib.load(), "rba_k2_variant")
    M, D, P = 8, 32, 4
    for name, hw in (("C5 3 levels", [(92, 160), (46, 80), (23, 40)]), ("C2 1 level", [(32, 64)])):
        L = len(hw)
        shapes = torch.tensor(hw, dtype=torch.int64).cuda()
        S = int(shapes.prod(1).sum())
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        g = torch.Generator().manual_seed(0)
        value = torch.randn(1, S, M, D, generator=g).cuda()
        refs = []
        for H_, W_ in hw:
            ry = (torch.arange(H_, dtype=torch.float32) + 0.5) / H_
            rx = (torch.arange(W_, dtype=torch.float32) + 0.5) / W_
            refs.append(torch.stack((rx[None, :].expand(H_, W_), ry[:, None].expand(H_, W_)), -1).reshape(-1, 2))
        ref = torch.cat(refs, 0)[None, :, None, :].repeat(1, 1, L, 1).contiguous().cuda()
        off = deform_ring_bias(M, L, P)[None, None] + 0.5 * torch.randn(1, S, M * L * P * 2, generator=g)
        raw = torch.cat([off, torch.randn(1, S, M * L * P, generator=g)], -1).contiguous().cuda()
        loc, w = ops.msda_prepare(raw, ref, shapes, M, L, P)
        alg = 4 * (S * M * D + 3 * S * M * L * P + S * M * D)
        rows = []
        variant.value = 1
        t_gen = timeit(lambda: ops.ms_deform_attn_forward(value, shapes, lsi, loc, w), reps)
        o_gen = ops.ms_deform_attn_forward(value, shapes, lsi, loc, w)
        variant.value = 0
        t_new = timeit(lambda: ops.ms_deform_attn_forward(value, shapes, lsi, loc, w), reps)
        t_prep = timeit(lambda: ops.msda_prepare(raw, ref, shapes, M, L, P), reps)
        t_fused = timeit(lambda: ops.msda_fused(value, shapes, lsi, raw, ref, M, L, P), reps)
        o_f = ops.msda_fused(value, shapes, lsi, raw, ref, M, L, P)
        print(f"{name}: {S} queries, algorithmic {alg / 1e6:.1f} MB | generic {t_gen:.1f} us ({alg / t_gen / 1e3 / 80:.1f} % of 8 TB/s)  round-3 {t_new:.1f} us "
              f"({alg / t_new / 1e3 / 80:.1f} %)  prepare {t_prep:.1f} us  prepare + round-3 {t_prep + t_new:.1f} us  fused {t_fused:.1f} us "
              f"({alg / t_fused / 1e3 / 80:.1f} %)  bit-identical {torch.equal(o_gen, o_f)}", flush=True)


if __name__ == "__main__":
    main()
