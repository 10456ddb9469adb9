#!/usr/bin/env python3
"""The row-complete token Linear (csrc/token_linear.hip) against what it replaced -- hipBLASLt GEMMs plus the element-wise / LayerNorm launches
between them -- on the encoder's and the decoder's small Linears, at C2's (2 048 tokens, 1 level) and C5's (4 830 tokens, 3 levels) token
counts.  Ten launches queued behind a long kernel per sample, median of five.      python tools/token_linear_ab.py"""
import os
import sys
from types import SimpleNamespace

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops

busy = torch.randn(8192, 8192, device="cuda")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        busy @ busy
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[2]


def lin(n, k, g):
    return SimpleNamespace(weight=(torch.randn(n, k, generator=g) * k ** -0.5).cuda(), bias=torch.randn(n, generator=g).cuda())


for M, L in ((2048, 1), (4830, 3)):
    g = torch.Generator().manual_seed(M)
    x, pos, res = (torch.randn(M, 256, generator=g).cuda() for _ in range(3))
    hid = torch.randn(M, 1024, generator=g).cuda()
    lv, ls, lo, l2 = lin(256, 256, g), lin(96 * L, 256, g), lin(256, 256, g), lin(256, 1024, g)
    norm = torch.nn.LayerNorm(256).cuda()
    parts = [(SimpleNamespace(weight=ls.weight[c:c + 256], bias=ls.bias[c:c + 256]), c) for c in range(0, 96 * L, 256)]
    raw = torch.empty(M, 96 * L, device="cuda")
    with torch.no_grad():
        t_new = timed(lambda: ops.token_linear_multi(x, [(lv, None, None, 0, False)] + [(pl, pos, raw, c0, False) for pl, c0 in parts]))
        t_old = timed(lambda: (F.linear(x, lv.weight, lv.bias), F.linear(x + pos, ls.weight, ls.bias)))
        print(f"M={M:5d}  value_proj + sampling Linear of (src + pos)   token kernel {t_new:6.1f} us   add + 2 library GEMMs {t_old:6.1f} us")
        t_new = timed(lambda: ops.token_linear(x, lo, residual=res, norm=norm))
        t_old = timed(lambda: ops.add_layer_norm(res, norm.weight, norm.bias, norm.eps, F.linear(x, lo.weight), lo.bias))
        print(f"M={M:5d}  output_proj + residual + LayerNorm (K = 256)    token kernel {t_new:6.1f} us   library GEMM + add_layer_norm {t_old:6.1f} us")
        t_new = timed(lambda: ops.token_linear(hid, l2, residual=res, norm=norm))
        t_old = timed(lambda: ops.add_layer_norm(res, norm.weight, norm.bias, norm.eps, F.linear(hid, l2.weight), l2.bias))
        print(f"M={M:5d}  linear2 + residual + LayerNorm (K = 1 024)      token kernel {t_new:6.1f} us   library GEMM + add_layer_norm {t_old:6.1f} us")
        t_new = timed(lambda: ops.token_linear_multi(x, [(lv, pos, None, 0, False), (lo, None, None, 0, False)]))
        t_old = timed(lambda: (F.linear(x + pos, lv.weight, lv.bias), F.linear(x, lo.weight, lo.bias)))
        print(f"M={M:5d}  decoder k = W_k (mem + pos), v = W_v mem         token kernel {t_new:6.1f} us   add + 2 library GEMMs {t_old:6.1f} us", flush=True)
