#!/usr/bin/env python3
"""Which op moves when test_non_finite_f16x3_score_is_rescored_on_bf16x6 sees the evaluator's re-scored map differ from the eager bf16x6 map (1 of ~10 fresh boxes)?
The test's sequence on the tiny net with every rba_amd.ops call (and F.conv2d / F.linear) of the bf16x6 forward of image 0 check-summed BEFORE and AFTER the evaluator
pass; prints the first op whose output bits differ.   python tools/flake_probe.py"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from rba_amd import arch as A, evaluate_ood as E, ops
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer
from rba_amd.support import OODEvaluator

ops.TILES_MIN = 1
a = A.complete(A.ARCHS["tiny3"])
model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
with torch.no_grad():
    model.backbone.layers[1].blocks[0].norm1.bias[3] = 1.0e5
g = torch.Generator().manual_seed(8)
imgs = [torch.randint(0, 256, (3, 128, 192), generator=g, dtype=torch.uint8) for _ in range(3)]
trace = None


def csum(o):
    if isinstance(o, torch.Tensor) and o.is_cuda and o.numel():
        t = o.detach().contiguous()
        if t.dtype in (torch.float32, torch.int32):
            return int(t.view(torch.int32).to(torch.int64).sum().item()) ^ (t.numel() << 40)
        return int(t.to(torch.int64).sum().item()) if not t.dtype.is_floating_point else float(t.double().sum().item())
    if isinstance(o, ops.SplitActivations):
        return csum(o.data)
    if isinstance(o, (tuple, list)):
        return tuple(csum(v) for v in o)
    return None


def wrap(mod, name):
    f = getattr(mod, name)

    def w(*args, **kw):
        out = f(*args, **kw)
        if trace is not None:
            trace.append((f"{mod.__name__.split('.')[-1]}.{name}", csum(out)))
        return out
    w.__wrapped_probe__ = True
    setattr(mod, name, w)


for n_, v_ in list(vars(ops).items()):
    if isinstance(v_, types.FunctionType) and not n_.startswith("_") and n_ not in ("split_mode", "set_concurrent_streams"):
        wrap(ops, n_)
for n_ in ("conv2d", "linear", "interpolate", "group_norm", "layer_norm"):
    wrap(F, n_)


def eager_bf16x6(k):
    global trace
    model.graph_replay = False
    trace = []
    with ops.split_mode("bf16x6"):
        out = model.rba_scores([{"image": imgs[k].cuda()}])[0].cpu().numpy()
    t, trace = trace, None
    return out, t


model.graph_replay = False
bad = model.rba_scores([{"image": imgs[0].cuda()}])[0]
print("f16x3 score finite:", bool(torch.isfinite(bad).all()))
before = [eager_bf16x6(k) for k in range(3)]
model.graph_replay = True
ev = OODEvaluator(model, E.get_logits, E.get_RbA)
gt = torch.zeros(1, 128, 192, dtype=torch.long)
gt[:, 10:40, 10:60] = 1
scores, gts = ev.compute_anomaly_scores([(im[None], gt) for im in imgs], device=torch.device("cuda"))
after = [eager_bf16x6(k) for k in range(3)]
again = [eager_bf16x6(k) for k in range(3)]
d = lambda x, y: (float(np.abs(x - y).max()), int((x != y).sum()))
for k in range(3):
    print(f"image {k}: evaluator vs before {d(scores[k], before[k][0])}  evaluator vs after {d(scores[k], after[k][0])}  before vs after {d(before[k][0], after[k][0])}  after vs again {d(after[k][0], again[k][0])}")
    tb, ta = before[k][1], after[k][1]
    if len(tb) != len(ta):
        print(f"  op counts differ: {len(tb)} vs {len(ta)}")
    for i, (x, y) in enumerate(zip(tb, ta)):
        if x != y:
            print(f"  first differing op: #{i} {x[0]} / {y[0]} (of {len(tb)} ops); the three before it: {[t[0] for t in tb[max(0, i - 3):i]]}")
            break
    else:
        print(f"  all {len(tb)} traced ops bit-equal before / after")

# ---- soak: the same traced bf16x6 forward of image 0 over and over (python tools/flake_probe.py N), every op's output check-summed: the first op that ever moves
N = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ref_out, ref_tr = eager_bf16x6(0)
moved = 0
for it in range(N):
    if it % 3 == 2:                                   # keep the captured f16x3 graphs and their streams busy in between, as the evaluator does
        model.graph_replay = True
        model.rba_scores([{"image": imgs[it % 3].cuda()}])
    out, tr = eager_bf16x6(0)
    if tr != ref_tr or not np.array_equal(out, ref_out):
        moved += 1
        for i, (x, y) in enumerate(zip(ref_tr, tr)):
            if x != y:
                print(f"forward {it}: first differing op #{i} {x[0]} (of {len(tr)}); before it {[t[0] for t in tr[max(0, i - 3):i]]}; final map {d(out, ref_out)}", flush=True)
                break
        if moved >= 5:
            break
print(f"soak: {N} traced forwards, {moved} differed from the first")
