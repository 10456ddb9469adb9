#!/usr/bin/env python3
"""Throughput probe: S concurrent HIP streams, each running the full Swin-B 1dl forward on its own image."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import arch as A
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

a = A.complete(A.ARCHS["swin_b_1dl"])
model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
imgs = [torch.randint(0, 256, (3, 1024, 2048), dtype=torch.uint8, device="cuda") for _ in range(4)]
for S in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(S)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                model.rba_scores([{"image": imgs[i % 4]}])
    run(2 * S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 12
    run(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{S} stream(s): {n / dt:.1f} images/s ({dt / n * 1e3:.2f} ms/image)")
# batch of 2 in one forward
t0 = time.perf_counter()
for i in range(6):
    model.rba_scores([{"image": imgs[0]}, {"image": imgs[1]}])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6):
    model.rba_scores([{"image": imgs[0]}, {"image": imgs[1]}])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"batch 2 per forward: {12 / dt:.1f} images/s")
