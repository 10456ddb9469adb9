"""When do K1's 1024 persistent workgroups start and finish?  (variant 110 of rba_reduce_f32_tune: wall-clock stamps per workgroup)
  python tools/k1_residency.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _tune

lib = _tune.load()
fn = lib.rba_reduce_f32_tune
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
Q, H, W = 100, 1024, 2048
g = torch.Generator(device="cuda").manual_seed(0)
mask = torch.randn(Q, H, W, device="cuda", generator=g) * 5
prob = torch.softmax(torch.randn(Q, 20, device="cuda", generator=g) * 3, -1)[:, :19].contiguous()
st = torch.cuda.current_stream().cuda_stream
for rnd in range(3):
    rba = torch.zeros(H, W, device="cuda")
    assert fn(mask.data_ptr(), prob.data_ptr(), rba.data_ptr(), Q, H * W, 110, st) == 0
    torch.cuda.synchronize()
    t = rba.view(-1).view(torch.int64)[: 2 * 1024].view(1024, 2).cpu().double()
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0           # 100 MHz -> us
    dur = end - start
    print(f"run {rnd}: kernel {end.max():.1f} us; workgroup start {start.min():.1f}..{start.max():.1f} us, end {end.min():.1f}..{end.max():.1f} us, "
          f"duration mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f}")
    for x in range(8):
        sel = torch.arange(1024) % 8 == x
        print(f"   XCD {x}: start {start[sel].mean():6.1f}  end mean {end[sel].mean():6.1f} max {end[sel].max():6.1f}  duration {dur[sel].mean():6.1f}")
    busy = (dur.sum() / 1024) / end.max()
    print(f"   mean residency {busy * 100:.1f} % of the kernel span")
