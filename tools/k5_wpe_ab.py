#!/usr/bin/env python3
"""K5 at 96 / 80 / 72 / 64 VGPRs (csrc/tune/k5_wpe_ab.hip): launch time on the four Swin-B stage shapes of a 1024x2048 image, warm and cold
operands, and bit-identity of the outputs."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _tune
from rba_amd import _lib, ops

lib = _tune.load()
fn = lib.rba_k5_wpe
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
fp = lib.rba_k5_wpe_plain
fp.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
fp.restype = ctypes.c_int
busy = torch.randn(8192, 8192, device="cuda")
big = torch.randn(96 << 20, device="cuda")                       # 384 MB: evicts the operands from the Infinity Cache (cold legs)
stages = [(256, 512, 4), (128, 256, 8), (64, 128, 16), (32, 64, 32)]
weights = (2, 2, 18, 2)
tot = {}
for (H, W, nH), wt in zip(stages, weights):
    C = nH * 32
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(1, H * W, 3 * C, device="cuda", generator=g)
    qb = torch.randn(3 * C, device="cuda", generator=g) * 0.1
    bias = torch.randn(nH, 144, 144, device="cuda", generator=g) * 0.5
    frag = ops.swin_bias_fragments(bias, 12)
    rows = (H * W + 31) // 32 * 32
    for shift in (0, 6):
        ref = None
        line = [f"{H}x{W} nH {nH} shift {shift}:"]
        for wpe in (5, 6, 7, 8, 15):                          # 15 = the ablation build's text at wpe 5 (72 VGPRs: its switches keep loads from being hoisted)
            for sout in (1,):
                out = torch.zeros(rows * C, device="cuda")
                st = torch.cuda.current_stream().cuda_stream

                def launch():
                    if wpe == 15:
                        _lib.check(fn(qkv.data_ptr(), qb.data_ptr(), frag.data_ptr(), out.data_ptr(), 1, H, W, nH, shift, sout, 5, 0, st), "k5 wpe")
                    else:
                        _lib.check(fp(qkv.data_ptr(), qb.data_ptr(), frag.data_ptr(), out.data_ptr(), 1, H, W, nH, shift, sout, wpe, st), "k5 wpe")
                launch()
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
                warm, cold = [], []
                for i in range(7):
                    busy @ busy
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        launch()
                    e1.record()
                    torch.cuda.synchronize()
                    if i >= 2:
                        warm.append(e0.elapsed_time(e1) * 1e2)
                for i in range(7):
                    big.add_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch()
                    e1.record()
                    torch.cuda.synchronize()
                    if i >= 2:
                        cold.append(e0.elapsed_time(e1) * 1e3)
                warm.sort(); cold.sort()
                w, c = warm[len(warm) // 2], cold[len(cold) // 2]
                tot[wpe] = tot.get(wpe, 0.0) + w * wt / 2
                tot[(wpe, "c")] = tot.get((wpe, "c"), 0.0) + c * wt / 2
                line.append(f"wpe{wpe} {w:6.1f} / {c:6.1f} us{'' if same else ' DIFFERS'}")
        print("  ".join(line))
print("per image (warm / cold single launches): " + "  ".join(f"wpe{w} {tot[w] / 1e3:.3f} / {tot[(w, 'c')] / 1e3:.3f} ms" for w in (5, 6, 7, 8, 15)))

# ablations (wrong results by construction except "heads fastest": launch time only); the ablation build's text has 72 VGPRs = two workgroups per CU at every wpe
for (H, W, nH) in ((64, 128, 16), (128, 256, 8), (256, 512, 4)):
    C = nH * 32
    qkv = torch.randn(1, H * W, 3 * C, device="cuda")
    qb = torch.randn(3 * C, device="cuda") * 0.1
    frag = ops.swin_bias_fragments(torch.randn(nH, 144, 144, device="cuda"), 12)
    out = torch.zeros((H * W + 31) // 32 * 32 * C, device="cuda")
    ref = None
    st = torch.cuda.current_stream().cuda_stream
    line = [f"{H}x{W} nH {nH} shift 0, warm / cold us:"]
    for ab, name in ((0, "product"), (0x4000, "heads fastest"), (1, "no bias loads"), (0x4001, "heads fastest, no bias"), (2, "no gather"), (4, "no stores"), (7, "no global traffic"),
                     (2 << 3, "late start 1 us")):
        def launch():
            _lib.check(fn(qkv.data_ptr(), qb.data_ptr(), frag.data_ptr(), out.data_ptr(), 1, H, W, nH, 0, 1, 5, ab, st), "k5 ablate")
        launch(); torch.cuda.synchronize()
        if ab == 0:
            ref = out.clone()
        tag = "" if ab not in (0, 0x4000) else (" same bits" if torch.equal(out.view(torch.int32), ref.view(torch.int32)) else " DIFFERS")
        ts, tc = [], []
        for i in range(7):
            busy @ busy
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                launch()
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1) * 1e2)
        for i in range(9):
            big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                tc.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); tc.sort()
        line.append(f"{name} {ts[len(ts) // 2]:.1f} / {tc[len(tc) // 2]:.1f}{tag}")
    print("  ".join(line))
