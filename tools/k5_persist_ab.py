#!/usr/bin/env python3
"""K5: persistent workgroups with the next item's gather in flight (csrc/tune/swin_window_attn_h3p.h, not adopted) against the product's one-shot kernel on the four Swin-B
stage shapes of a 1024x2048 image (and Swin-L's with `l`): warm (10 queued launches) and cold (operands evicted from the Infinity Cache) launch times."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _tune
from rba_amd import _lib, ops

fnp = _tune.load().rba_k5_persist
fnp.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
fnp.restype = ctypes.c_int
busy = torch.randn(8192, 8192, device="cuda")
big = torch.randn(96 << 20, device="cuda")
large = len(sys.argv) > 1 and sys.argv[1] == "l"
stages = [(256, 512, 6), (128, 256, 12), (64, 128, 24), (32, 64, 48)] if large else [(256, 512, 4), (128, 256, 8), (64, 128, 16), (32, 64, 32)]
tot = {}
for (H, W, nH), wt in zip(stages, (2, 2, 18, 2)):
    C = nH * 32
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(1, H * W, 3 * C, device="cuda", generator=g)
    qb = torch.randn(3 * C, device="cuda", generator=g) * 0.1
    bias = torch.randn(nH, 144, 144, device="cuda", generator=g) * 0.5
    frag = ops.swin_bias_fragments(bias, 12)
    for shift in (0, 6):
        line = [f"{H}x{W} nH {nH} shift {shift}:"]
        ref = None
        for v in (1, 0):
            pout = ops.SplitActivations.empty((1, H * W, C), qkv.device)

            def run():
                if v:
                    return ops.swin_window_attn(qkv, qb, bias, H, W, nH, 12, shift, bias_frag=frag, split_out=True)
                _lib.check(fnp(qkv.data_ptr(), qb.data_ptr(), frag.data_ptr(), pout.data.data_ptr(), 1, H, W, nH, shift, 1, torch.cuda.current_stream().cuda_stream), "k5 persist")
                return pout
            try:
                out = run()
                torch.cuda.synchronize()
                nfull = (H * W) // 32 * 32 * C
                same = ref is None or torch.equal(out.data[:nfull].view(torch.int32), ref[:nfull].view(torch.int32))
                if ref is None:
                    ref = out.data.clone()
                warm, cold = [], []
                for i in range(7):
                    busy @ busy
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    if i >= 2:
                        warm.append(e0.elapsed_time(e1) * 1e2)
                for i in range(9):
                    big.add_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run()
                    e1.record()
                    torch.cuda.synchronize()
                    if i >= 2:
                        cold.append(e0.elapsed_time(e1) * 1e3)
            finally:
                pass
            warm.sort(); cold.sort()
            w, c = warm[len(warm) // 2], cold[len(cold) // 2]
            tot[v] = tot.get(v, 0.0) + w * wt / 2
            tot[(v, "c")] = tot.get((v, "c"), 0.0) + c * wt / 2
            line.append(f"{'one-shot' if v else 'persistent'} {w:6.1f} / {c:6.1f} us{'' if same else ' DIFFERS'}")
        print("  ".join(line))
print("per image, warm / cold: " + "  ".join(f"{'one-shot' if v else 'persistent'} {tot[v] / 1e3:.3f} / {tot[(v, 'c')] / 1e3:.3f} ms" for v in (1, 0)))
