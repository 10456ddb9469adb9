"""Run one split-linear shape repeatedly (for rocprofv3 counter passes):  python tools/gemm_one.py M N K [iters] [v4 cfg | hCFG]
(cfg "h1004": the f16x3 tune entry with configuration 1004, see tools/gemm_h3_sweep.py)"""
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
arg = sys.argv[5] if len(sys.argv) > 5 else "0"
if arg.startswith("h"):
    import ctypes
    import _tune
    from rba_amd import _lib
    fn = _tune.load().rba_split_linear_h3_tune
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    planes = ops.split_weight(w, mode="f16x3")
    y = torch.empty(M, N, device="cuda")
    for _ in range(it):
        _lib.check(fn(x.data_ptr(), planes.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, 1, int(arg[1:]), torch.cuda.current_stream().cuda_stream), "h3")
    torch.cuda.synchronize()
    print("ok", float(y[0, 0]))
    sys.exit(0)
cfg = int(arg)
planes = ops.split_weight(w, mode="bf16x6" if cfg else None)
for _ in range(it):
    y = split_linear_cfg(x, planes, b, cfg=cfg) if cfg else ops.split_linear(x, planes, b)
torch.cuda.synchronize()
print("ok", float(y[0, 0]))
