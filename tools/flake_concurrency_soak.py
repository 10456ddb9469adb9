#!/usr/bin/env python3
"""Hypothesis for the rare last-bit difference of the bf16x6 re-score (tests/test_model_gpu.py::test_non_finite_f16x3_score_is_rescored_on_bf16x6): the evaluator
re-scores image 0 while the f16x3 graph replays of the following images may still be in flight.  N iterations of: replay images 1 and 2 (captured graphs, no
synchronisation), then the eager bf16x6 forward of image 0, compared bit for bit with the first one.   python tools/flake_concurrency_soak.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rba_amd import arch as A, ops
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ops.TILES_MIN = 1
a = A.complete(A.ARCHS["tiny3"])
model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
with torch.no_grad():
    model.backbone.layers[1].blocks[0].norm1.bias[3] = 1.0e5
g = torch.Generator().manual_seed(8)
imgs = [torch.randint(0, 256, (3, 128, 192), generator=g, dtype=torch.uint8).cuda() for _ in range(3)]


def eager(k):
    model.graph_replay = False
    with ops.split_mode("bf16x6"):
        return model.rba_scores([{"image": imgs[k]}])[0].clone()


model.graph_replay = True
for _ in range(4):
    for k in range(3):
        model.rba_scores([{"image": imgs[k]}])           # second occurrence captures, later ones replay
print("live graphs:", model.live_graphs())
ref = eager(0)
torch.cuda.synchronize()
bad = 0
for it in range(N):
    model.graph_replay = True
    model.rba_scores([{"image": imgs[1]}])
    model.rba_scores([{"image": imgs[2]}])
    out = eager(0)
    if not torch.equal(out, ref):
        bad += 1
        d = (out - ref).abs()
        print(f"iteration {it}: {int((out != ref).sum())} pixels differ, max {float(d.max()):.3e}, nan {int(torch.isnan(out).sum())}", flush=True)
        if bad >= 10:
            break
print(f"{N} iterations of (two graph replays in flight, eager bf16x6 forward): {bad} differed")
