#!/usr/bin/env python3
"""Host-side cost of one forward: time to ISSUE the launches of model.rba_scores (no synchronisation inside the loop) against the
GPU time of the same calls, with and without decode-like threads competing for the interpreter lock."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rba_amd import arch as A
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

a = A.complete(A.ARCHS["swin_b_1dl"])
m = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
img = torch.randint(0, 256, (3, 1024, 2048), dtype=torch.uint8).cuda()
for _ in range(3):
    m.rba_scores([{"image": img}])
torch.cuda.synchronize()


def run(n=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.rba_scores([{"image": img}])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


print("issue %.2f ms / image, issue + drain %.2f ms / image (no other threads)" % run())
stop = False


def busy():                       # numpy <-> torch conversions like the decode threads' non-PIL part
    x = np.random.randint(0, 255, (1024, 2048, 3), dtype=np.uint8)
    while not stop:
        torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))).clone()


for nthreads in (2, 8):
    stop = False
    ts = [threading.Thread(target=busy, daemon=True) for _ in range(nthreads)]
    [t.start() for t in ts]
    print("issue %.2f ms / image, issue + drain %.2f ms / image" % run(), f"with {nthreads} conversion threads")
    stop = True
    [t.join() for t in ts]
