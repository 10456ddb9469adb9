#!/usr/bin/env python3
"""A few launches of K5 on one stage shape (for counter passes): python tools/k5_one.py [H W nH shift iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rba_amd import ops

H, W, nH, shift, iters = (int(v) for v in (sys.argv[1:6] + ["64", "128", "16", "0", "5"][len(sys.argv) - 1:]))
C = nH * 32
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(1, H * W, 3 * C, device="cuda", generator=g)
qb = torch.randn(3 * C, device="cuda", generator=g) * 0.1
bias = torch.randn(nH, 144, 144, device="cuda", generator=g) * 0.5
frag = ops.swin_bias_fragments(bias, 12)
for _ in range(iters):
    out = ops.swin_window_attn(qkv, qb, bias, H, W, nH, 12, shift, bias_frag=frag, split_out=True)
torch.cuda.synchronize()
