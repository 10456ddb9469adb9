"""Split-bf16 (bf16x6) Linear vs hipBLASLt fp32 on the Swin token-GEMM shapes: accuracy against fp64 and event-timed speed.
  python tools/gemm_sweep.py [swin_b|swin_l|c5]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from rba_amd import ops


def shapes(kind):
    if kind == "swin_l":
        C, T = 192, [131072, 32768, 8192, 2048]
    elif kind == "c5":
        C, T = 128, [57600, 14400, 3600, 900]
    else:
        C, T = 128, [131072, 32768, 8192, 2048]
    out = []
    for s, t in enumerate(T):
        c = C << s
        out += [(f"s{s+1} qkv", t, 3 * c, c), (f"s{s+1} proj", t, c, c), (f"s{s+1} fc1", t, 4 * c, c), (f"s{s+1} fc2", t, c, 4 * c)]
    return out


def timeit(fn, n=20):
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


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
    torch.manual_seed(0)
    dev = "cuda"
    tot_a = tot_b = 0.0
    for name, M, N, K in shapes(kind):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * (K ** -0.5)
        b = torch.randn(N, device=dev)
        if not ops.split_linear_supported(N, K):
            print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  unsupported")
            continue
        planes = ops.split_weight(w)
        assert torch.equal(ops.unpack_split_weight(planes)[:, :N].float().sum(0), w), "planes do not sum to the weight"
        y = ops.split_linear(x, planes, b, out_features=N)
        y0 = F.linear(x, w, b)
        rows = torch.randperm(M, device=dev)[:512]
        ref = (x[rows].double() @ w.double().T + b.double())
        e_new = (y[rows].double() - ref).abs().max().item()
        e_old = (y0[rows].double() - ref).abs().max().item()
        t_old = timeit(lambda: F.linear(x, w, b))
        t_new = timeit(lambda: ops.split_linear(x, planes, b, out_features=N))
        fl = 2.0 * M * N * K
        tot_a += t_old
        tot_b += t_new
        print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  hipBLASLt {t_old:7.1f} us ({fl/t_old*1e-6:6.1f} TF)  bf16x6 {t_new:7.1f} us "
              f"({fl/t_new*1e-6:6.1f} TF)  x{t_old/t_new:4.2f}   err vs fp64: fp32 {e_old:.2e}  bf16x6 {e_new:.2e}", flush=True)
    print(f"sum: hipBLASLt {tot_a:.0f} us, bf16x6 {tot_b:.0f} us")
    # GELU epilogue
    x = torch.randn(8192, 512, device=dev); w = torch.randn(2048, 512, device=dev) * 512 ** -0.5; b = torch.randn(2048, device=dev)
    planes = ops.split_weight(w)
    yg = ops.split_linear(x, planes, b, gelu=True)
    print("gelu epilogue max|d| vs F.gelu(F.linear):", (yg - F.gelu(F.linear(x, w, b))).abs().max().item(),
          " time", timeit(lambda: ops.split_linear(x, planes, b, gelu=True)), "us vs", timeit(lambda: F.gelu(F.linear(x, w, b))), "us")


if __name__ == "__main__":
    main()
