"""K6 v4 (all-LDS-DMA bf16x6 Linear, split_linear_dma.hip) tile-configuration sweep against the round-1 pipe kernel on the Swin
token-GEMM shapes: interleaved rounds in one process, median of event-timed launches, accuracy against fp64.
  python tools/gemm_v4_sweep.py [swin_b|swin_l|c5] [cfg,cfg,...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from _tune import ops, split_linear_cfg
from gemm_sweep import shapes

CFGS = [1421, 1412, 1413, 1221, 1222, 1612, 2411]


def lds_ok(cfg, N):
    return True


def bench_round(fns, n=10):
    """one interleaved round: each fn timed over n back-to-back launches -> us per launch"""
    out = []
    for fn in fns:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
    cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else CFGS
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    torch.manual_seed(0)
    dev = "cuda"
    best_sum = base_sum = 0.0
    for name, M, N, K in shapes(kind):
        if only and not any(o in name for o in only):
            continue
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * (K ** -0.5)
        b = torch.randn(N, device=dev)
        planes = ops.split_weight(w)
        rows = torch.randperm(M, device=dev)[:512]
        ref = (x[rows].double() @ w.double().T + b.double())
        act = 1 if "fc1" in name else 0
        y0 = ops.split_linear(x, planes, b, gelu=bool(act), out_features=N)
        fns = [lambda: ops.split_linear(x, planes, b, gelu=bool(act), out_features=N)]
        names = ["r1"]
        errs = {}
        for c in cfgs:
            try:
                y = split_linear_cfg(x, planes, b, act=act, cfg=c, out_features=N)
            except Exception as e:
                print(f"  cfg {c}: {e}")
                continue
            torch.cuda.synchronize()
            errs[c] = (y - y0).abs().max().item()
            if act == 0:
                errs[c] = max(errs[c], (y[rows].double() - ref).abs().max().item())
            fns.append(lambda c=c: split_linear_cfg(x, planes, b, act=act, cfg=c, out_features=N))
            names.append(str(c))
        rounds = [bench_round(fns) for _ in range(5)]
        med = [sorted(r[i] for r in rounds)[len(rounds) // 2] for i in range(len(fns))]
        fl = 2.0 * M * N * K
        bi = min(range(1, len(med)), key=lambda i: med[i]) if len(med) > 1 else 0
        base_sum += med[0]
        best_sum += min(med)
        print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d} act={act} " + "  ".join(f"{n}:{t:6.1f}" for n, t in zip(names, med)) +
              f"  | best {names[bi]} {fl/med[bi]*1e-6:6.1f} TF ({6*fl/med[bi]*1e-6/2500*100:4.1f}% of bf16 peak)  max err {max(errs.values()) if errs else 0:.1e}",
              flush=True)
    print(f"sum: r1 {base_sum:.0f} us, best-of {best_sum:.0f} us")


if __name__ == "__main__":
    main()
