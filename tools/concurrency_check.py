#!/usr/bin/env python3
"""Are the kernels bitwise reproducible while ANOTHER process uses the same GPU?  Spawns N copies of itself; each repeats (a) the
K6 GEMM in both forms and (b) the whole forward on fixed inputs and counts results that differ from its first one.
  python tools/concurrency_check.py [nproc] [iters] [arch H W]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if os.environ.get("CC_CHILD"):
    import torch
    from rba_amd import arch as A, ops
    from rba_amd.checkpoint import load_checkpoint
    from rba_amd.maskformer_model import MaskFormer
    iters = int(sys.argv[2])
    name, h, w = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    torch.manual_seed(0)
    x = torch.randn(8192, 512, device="cuda")
    wt = torch.randn(2048, 512, device="cuda") * 512 ** -0.5
    b = torch.randn(2048, device="cuda")
    bad = {}
    for mode in ("f16x3", "bf16x6"):
        p = ops.split_weight(wt, mode=mode)
        first = ops.split_linear(x, p, b, gelu=True).clone()
        bad[mode] = sum(int(not torch.equal(ops.split_linear(x, p, b, gelu=True), first)) for _ in range(iters * 20))
    a = A.complete(A.ARCHS[name])
    m = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
    img = torch.randint(0, 256, (3, h, w), generator=torch.Generator().manual_seed(3), dtype=torch.uint8).cuda()
    with torch.no_grad():
        first = m.rba_scores([{"image": img}])[0].clone()
        nbad, worst = 0, 0.0
        for _ in range(iters):
            r = m.rba_scores([{"image": img}])[0]
            if not torch.equal(r, first):
                nbad += 1
                worst = max(worst, float((r - first).abs().max()))
    print(f"child {os.environ['CC_CHILD']}: GEMM results differing {bad} of {iters * 20}; forwards differing {nbad}/{iters} (max |d| {worst:.2e}); "
          f"checksum {float(first.double().sum()):.10f}", flush=True)
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = sys.argv[2] if len(sys.argv) > 2 else "20"
    rest = sys.argv[3:6] if len(sys.argv) > 5 else ["swin_b_1dl", "512", "1024"]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(n), iters] + rest, env=dict(os.environ, CC_CHILD=str(i)))
             for i in range(n)]
    sys.exit(max(p.wait() for p in procs))
