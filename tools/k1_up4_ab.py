#!/usr/bin/env python3
"""K1 fused with the x4 upsample (rba_reduce_up4_f32: the product's default K1 path): the generic kernel (rba_k1_up4_variant = 1), the packed VALU kernel
(2) and -- score only -- the matrix-pipe kernel (0 = product dispatch), at 1024 x 2048 and 720 x 1280; ten launches queued behind a long kernel per sample.
python tools/k1_up4_ab.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import _lib, ops

var = ctypes.c_int.in_dll(_lib.load(), "rba_k1_up4_variant")
busy = torch.randn(8192, 8192, device="cuda")
for (H, W), crop in (((1024, 2048), (1024, 2048)), ((736, 1280), (720, 1280))):
    g = torch.Generator().manual_seed(0)
    low = (torch.randn(100, H // 4, W // 4, generator=g) * 5).cuda()
    prob = torch.softmax(torch.randn(100, 20, generator=g) * 3, -1)[:, :19].contiguous().cuda()
    for full in (False, True):
        res = {}
        for v in (1, 2, 0):
            var.value = v
            for _ in range(3):
                out = ops.rba_reduce_up4(low, prob, crop, full, full)
            ts = []
            for _ in range(5):
                busy @ busy
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    out = ops.rba_reduce_up4(low, prob, crop, full, full)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e2)
            ts.sort()
            res[v] = (ts[len(ts) // 2], out)
        var.value = 0
        d = (res[2][1][0] - res[1][1][0]).abs().max().item()
        dm = (res[0][1][0] - res[1][1][0]).abs().max().item()
        print(f"{crop} {'score + sem_seg + argmax' if full else 'score only':26s} generic {res[1][0]:7.1f} us  packed {res[2][0]:7.1f} us (max|d rba| {d:.1e})  "
              f"product dispatch {res[0][0]:7.1f} us (max|d rba| {dm:.1e})" + (f"  argmax equal {torch.equal(res[0][1][2], res[1][1][2])}" if full else ""), flush=True)
