#!/usr/bin/env python3
"""Soak of the product's GroupNorm-folded projection (csrc/split_linear_gnf.hip): 2 x 100 launches of the full-size case, every one bit-identical to the two-call form
(round 5: 0 / 200 differ; the builds with the packed cross-select multiply differed in EVERY launch, profiles/r05_gnfold_select.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops
g = torch.Generator().manual_seed(7)
B, P, K, N, G = 1, 131072, 256, 256, 32
x = (torch.randn(B, P, K, generator=g) * 3 + 0.7).cuda()
w, b = (torch.randn(N, K, generator=g) * K ** -0.5).cuda(), torch.randn(N, generator=g).cuda()
ga, be = (torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda()
p3 = ops.split_weight(w, mode="f16x3")
for relu in (True, False):
    two = ops.split_linear_nchw_out(ops.group_norm_nhwc(x, G, ga, be, 1e-5, relu=relu).view(B * P, K), p3, b, P, out_features=N)
    mr = ops.group_norm_nhwc_stats(x, G, 1e-5)
    bad = 0
    for i in range(100):
        one = ops.split_linear_nchw_out_gn(x.view(B * P, K), mr, ga, be, G, relu, p3, b, P, out_features=N)
        bad += int(not torch.equal(one, two))
    print("relu", relu, "launches 100, not bit-identical to the two-call form:", bad)
