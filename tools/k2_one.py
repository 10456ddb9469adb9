#!/usr/bin/env python3
"""A few launches of K2 at BASELINE C5's encoder shape for the profiler (tools/pmc_k2.sh): `python tools/k2_one.py [fused|fwd|generic] [reps]`"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import _lib, ops  # noqa: E402
from rba_amd.seeded_weights import deform_ring_bias  # noqa: E402

form = sys.argv[1] if len(sys.argv) > 1 else "fused"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M, D, P = 8, 32, 4
hw = [(92, 160), (46, 80), (23, 40)]
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
if form == "generic":
    ctypes.c_int.in_dll(_lib.load(), "rba_k2_variant").value = 1
for _ in range(reps):
    if form == "fused":
        ops.msda_fused(value, shapes, lsi, raw, ref, M, L, P)
    else:
        ops.ms_deform_attn_forward(value, shapes, lsi, loc, w)
torch.cuda.synchronize()
