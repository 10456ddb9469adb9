#!/usr/bin/env python3
"""Where one image's GPU time goes, op by op: every rba_amd.ops launch of one eager forward (one stream) between two HIP events, grouped by
(op, shapes of its tensor arguments) in first-call order -- the per-stage view the kernel trace's by-name table cannot give (the same K6
instantiation serves several stages).  Times include the events' own few microseconds.

  python tools/stage_times.py [arch] [H W] [iters]      -> a table on stdout"""
import os
import sys
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from rba_amd import arch as A, ops
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

name = sys.argv[1] if len(sys.argv) > 1 else "swin_b_1dl"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 2048)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 8
a = A.complete(A.ARCHS[name])
model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
model.graph_replay = False
g = torch.Generator().manual_seed(5)
image = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).cuda()
state = {"on": False, "depth": 0}
calls = {}
order = []


def shapes(args):
    out = []
    for v in args:
        if isinstance(v, torch.Tensor):
            out.append("x".join(map(str, v.shape)) or "s")
        elif isinstance(v, ops.SplitActivations):
            out.append("split[" + "x".join(map(str, getattr(v, "shape", v.data.shape))) + "]")
        elif isinstance(v, torch.nn.Linear):
            out.append(f"Linear({v.in_features}->{v.out_features})")
        elif isinstance(v, torch.nn.Conv2d):
            out.append(f"Conv({v.in_channels}->{v.out_channels},k{v.kernel_size[0]})")
    return ",".join(out[:4])


def wrap(fn, label):
    def inner(*args, **kw):
        if not state["on"] or state["depth"]:
            return fn(*args, **kw)
        state["depth"] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            r = fn(*args, **kw)
        finally:
            state["depth"] -= 1
        e1.record()
        key = (label, shapes(list(args) + list(kw.values())))
        if key not in calls:
            calls[key] = []
            order.append(key)
        calls[key].append((e0, e1))
        return r
    return inner


SKIP = {"split_mode", "split_weight", "linear_takes_split", "split_linear_pays", "split_linear_supported", "mlp_fused_ok", "linear_residual_fused",
        "token_linear_ok", "token_linear_pays", "msda_fused_ok", "set_concurrent_streams", "swin_window_attn_split_ok"}
for k in dir(ops):
    v = getattr(ops, k)
    if isinstance(v, types.FunctionType) and not k.startswith("_") and k not in SKIP and v.__module__ == ops.__name__:
        setattr(ops, k, wrap(v, k))

with torch.no_grad():
    for _ in range(3):
        model.rba_scores([{"image": image}])
    torch.cuda.synchronize()
    state["on"] = True
    tot = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.rba_scores([{"image": image}])
        e1.record()
        tot.append((e0, e1))
    torch.cuda.synchronize()
total = sorted(s.elapsed_time(e) for s, e in tot)[len(tot) // 2] * 1e3
rows, covered = [], 0.0
for key in order:
    v = [s.elapsed_time(e) * 1e3 for s, e in calls[key]]
    per = len(v) // iters
    us = sum(v) / iters
    covered += us
    rows.append((key, per, us))
print(f"# {name} {H}x{W}: eager launches, one stream, mean of {iters} forwards; whole rba_scores {total:.0f} us (host-paced), inside rba_amd.ops launches {covered:.0f} us")
print("| op | tensor arguments | calls | us per image | us per call |")
print("|---|---|---:|---:|---:|")
for (label, shp), per, us in rows:
    print(f"| {label} | {shp} | {per} | {us:.0f} | {us / max(per, 1):.1f} |")
