"""bf16x6 implicit-GEMM 3x3 convolution (NHWC) vs MIOpen (F.conv2d, NCHW) on the FPN output-conv shapes:  python tools/conv_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from rba_amd import ops


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
for H, W in ((256, 512), (128, 256), (64, 128), (180, 320), (90, 160)):
    C = N = 256
    x = torch.randn(1, C, H, W, device="cuda")
    w = torch.randn(N, C, 3, 3, device="cuda") * (9 * C) ** -0.5
    xn = x.permute(0, 2, 3, 1).contiguous()
    planes = ops.conv3x3_weight(w)
    y = ops.conv3x3_nhwc(xn, planes)
    y0 = F.conv2d(x, w, padding=1)
    ref = F.conv2d(x[:, :, :40].double(), w.double(), padding=1)[:, :, :32].permute(0, 2, 3, 1)
    e_new = (y[:, :32].double() - ref).abs().max().item()
    e_old = (y0[:, :, :32].double().permute(0, 2, 3, 1) - ref).abs().max().item()
    t_old = timeit(lambda: F.conv2d(x, w, padding=1))
    t_new = timeit(lambda: ops.conv3x3_nhwc(xn, planes))
    fl = 2.0 * H * W * N * 9 * C
    print(f"{H:4d}x{W:4d}x{C}: MIOpen {t_old:7.1f} us ({fl/t_old*1e-6:6.1f} TF)  bf16x6 implicit GEMM {t_new:7.1f} us ({fl/t_new*1e-6:6.1f} TF)  "
          f"x{t_old/t_new:4.2f}   err vs fp64: MIOpen {e_old:.2e}  bf16x6 {e_new:.2e}", flush=True)
