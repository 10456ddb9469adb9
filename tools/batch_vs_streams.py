#!/usr/bin/env python3
"""One forward over a BATCH of B images on one stream (larger launches: whole rounds of tiles, weights amortised) against B = 1 -- the alternative to
bench.py's "each image of a step on its own stream".  GPU time per image up to the decoder heads (model.predict), eager launches, events around 10 forwards.
Usage: tools/batch_vs_streams.py [arch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import arch as A
from rba_amd import ops
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

arch = sys.argv[1] if len(sys.argv) > 1 else "swin_b_1dl"
dev = torch.device("cuda", 0)
a = A.complete(A.ARCHS[arch])
model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).to(dev).eval()
imgs = [torch.randint(0, 256, (3, 1024, 2048), generator=torch.Generator().manual_seed(i), dtype=torch.uint8).to(dev) for i in range(4)]

for hint in (1, 3):
    ops.set_concurrent_streams(hint)
    for B in (1, 2, 3, 4):
        batch = [{"image": imgs[i]} for i in range(B)]
        with torch.no_grad():
            for _ in range(3):
                model.predict(batch)
            torch.cuda.synchronize()
            ts = []
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    model.predict(batch)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 / B)
        ts.sort()
        print(f"{arch} launch-geometry hint {hint}  B={B}: {ts[1]:.3f} ms per image  ({1e3 / ts[1]:.1f} images/s up to the decoder heads)", flush=True)

# ---- S concurrent streams x batches of B, each stream replaying a captured graph of model.predict (bench.py's multi-stream mechanism without the K1 part)
print("streams x batch, hipGraph replay per stream:", flush=True)
for (S, B) in ((3, 1), (1, 4), (2, 2), (3, 2), (2, 3), (2, 4), (4, 2), (1, 8), (2, 6)):
    ops.set_concurrent_streams(S)
    graphs = []
    with torch.no_grad():
        for j in range(S):
            st = torch.cuda.Stream()
            batch = [{"image": imgs[(j + i) % 4].clone()} for i in range(B)]
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                model.predict(batch)
            torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                out = model.predict(batch)
            torch.cuda.synchronize()
            graphs.append((g, st, out, batch))
    main = torch.cuda.current_stream()

    def step():
        for g, st, _, _ in graphs:
            st.wait_stream(main)
            with torch.cuda.stream(st):
                g.replay()
        for g, st, _, _ in graphs:
            main.wait_stream(st)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 / (S * B))
    ts.sort()
    print(f"  S={S} B={B}: {ts[1]:.3f} ms per image  ({1e3 / ts[1]:.1f} images/s up to the decoder heads)", flush=True)
    del graphs
