#!/usr/bin/env python3
"""How much slower do the hot kernels run on cold caches?  Each kernel is timed back to back (operands warm in L2 / Infinity Cache) and right
after a 1.3 GB copy that sweeps both (the copy's own time subtracted); ten launches queued behind a long kernel per sample.  python tools/cold_probe.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops
busy = torch.randn(8192, 8192, device="cuda")
thr_src = torch.randn(160 * 1024 * 1024, device="cuda")       # 640 MB: beyond L2 + Infinity Cache
thr_dst = torch.empty_like(thr_src)
g = torch.Generator().manual_seed(0)
def timed(fns, reps=8):
    ts = []
    for _ in range(5):
        busy @ busy
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for f in fns: f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort(); return ts[2]
thrash = lambda: thr_dst.copy_(thr_src)
t_thr = timed([thrash])
print(f"thrash alone {t_thr:.1f} us")
def case(name, make):
    fn_same, fn_cold_w = make()
    warm = timed([fn_same], 10)
    cold_all = timed([thrash, fn_same]) - t_thr
    print(f"{name:34s} warm (back to back) {warm:6.1f} us | after a 1.3 GB sweep of the caches {cold_all:6.1f} us", flush=True)
def gemm(M, N, K, mode):
    def make():
        x = torch.randn(M, K, generator=g).cuda(); lin = torch.nn.Linear(K, N).cuda(); r = torch.randn(M, N, generator=g).cuda()
        xs = ops.SplitActivations.pack(x)
        if mode == "res": f = lambda: ops.linear(xs, lin, residual=r)
        elif mode == "gelu": f = lambda: ops.linear(xs, lin, gelu=True, split_out=True)
        else: f = lambda: ops.linear(xs, lin)
        for _ in range(3): f()
        return f, None
    return make
case("s3 qkv 8192x1536x512", gemm(8192, 1536, 512, "f32"))
case("s3 proj+res 8192x512x512", gemm(8192, 512, 512, "res"))
case("s3 fc1+gelu 8192x2048x512", gemm(8192, 2048, 512, "gelu"))
case("s3 fc2+res 8192x512x2048", gemm(8192, 512, 2048, "res"))
case("c5 s3 fc2+res 3680x512x2048", gemm(3680, 512, 2048, "res"))
def k5():
    H, W, nH, ws = 64, 128, 16, 12
    C = nH * 32
    qkv = torch.randn(1, H * W, 3 * C, device="cuda"); qb = torch.randn(3 * C, device="cuda") * 0.1
    bias = torch.randn(nH, 144, 144, device="cuda") * 0.5; frag = ops.swin_bias_fragments(bias, ws)
    f = lambda: ops.swin_window_attn(qkv, qb, bias, H, W, nH, ws, 6, bias_frag=frag)
    for _ in range(3): f()
    return f, None
case("K5 stage 3 (64x128, 16 heads)", k5)
def ln():
    x = torch.randn(8192, 512, device="cuda"); w = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda")
    f = lambda: ops.add_layer_norm(x, w, b, 1e-5, frag=True)
    for _ in range(3): f()
    return f, None
case("LayerNorm 8192x512 -> split image", ln)
