"""Run one split-linear shape repeatedly (for rocprofv3 counter passes):  python tools/gemm_one.py M N K [iters] [v4 cfg]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from _tune import ops, split_linear_cfg

M, N, K = (int(v) for v in sys.argv[1:4])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * K ** -0.5
b = torch.randn(N, device="cuda")
planes = ops.split_weight(w)
cfg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
for _ in range(it):
    y = split_linear_cfg(x, planes, b, cfg=cfg) if cfg else ops.split_linear(x, planes, b)
torch.cuda.synchronize()
print("ok", float(y[0, 0]))
