"""Event-time split_linear on a few shapes (kernel variant from RBA_GEMM_VARIANT):  python tools/gemm_time.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from rba_amd import ops

SHAPES = [(8192, 512, 2048), (8192, 2048, 512), (32768, 256, 1024), (8192, 1536, 512)]
torch.manual_seed(0)
out = []
for M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda")
    planes = ops.split_weight(torch.randn(N, K, device="cuda") * K ** -0.5)
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        ops.split_linear(x, planes, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.split_linear(x, planes, b)
    e1.record()
    torch.cuda.synchronize()
    out.append(f"{e0.elapsed_time(e1) / 20 * 1e3:7.1f}")
print(f"variant {os.environ.get('RBA_GEMM_VARIANT', 'default'):>8s}: " + " ".join(out) + "  us for " + " ".join(f"{m}x{n}x{k}" for m, n, k in SHAPES))
