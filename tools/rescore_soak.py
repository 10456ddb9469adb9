#!/usr/bin/env python3
"""Soak of the scenario of tests/test_model_gpu.py::test_non_finite_f16x3_score_is_rescored_on_bf16x6, the one comparison that was seen to
differ once in round 3: the tiny 3-level model with every Linear on K6 (ops.TILES_MIN = 1: launches of 1-12 tiles, K = 32 ... 256 -- shapes the
product never runs), one LayerNorm bias poisoned to 1e5, scored eagerly in the bf16x6 arithmetic; and the same model un-poisoned in f16x3.
Every op of every forward is compared with the first forward's (bitwise-equal inputs, different output = that op is not reproducible), with
the caching allocator perturbed between forwards and a K6 GEMM loop on another stream.

  python tools/rescore_soak.py [forwards]      -> one JSON line per arithmetic mode"""
import json
import os
import sys
import threading
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from rba_amd import arch as A, ops
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ops.TILES_MIN = 1
state = {"mode": "off", "pos": 0, "log": [], "report": {}}


def tens(args):
    out = []
    for v in args:
        if isinstance(v, torch.Tensor):
            out.append(v)
        elif isinstance(v, ops.SplitActivations):
            out.append(v.data)
        elif isinstance(v, (list, tuple)):
            out += tens(v)
        elif isinstance(v, torch.nn.Module):
            out += list(v.parameters())
    return out


def wrap(fn, label):
    def inner(*args, **kw):
        r = fn(*args, **kw)
        if state["mode"] == "off":
            return r
        ins = [t.detach().clone() for t in tens(args) + tens(list(kw.values()))]
        outs = [t.detach().clone() for t in tens([r])]
        if state["mode"] == "record":
            state["log"].append((label, ins, outs))
        else:
            lab, i0, o0 = state["log"][state["pos"]]
            assert lab == label, (lab, label)
            same_in = len(i0) == len(ins) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(i0, ins))
            same_out = all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(o0, outs))
            if same_in and not same_out:
                key = (state["pos"], label)
                d = max(float((x.float() - y.float()).abs().max()) for x, y in zip(o0, outs) if x.dtype.is_floating_point) if outs else 0.0
                e = state["report"].setdefault(key, {"op": state["pos"], "label": label, "input_shapes": [tuple(t.shape) for t in ins][:4], "times": 0, "max_abs_diff": 0.0})
                e["times"] += 1
                e["max_abs_diff"] = max(e["max_abs_diff"], d)
            state["pos"] += 1
        return r
    return inner


_split_linear, _split_weight = ops.split_linear, ops.split_weight          # unwrapped: the GEMM thread beside the soak must not touch the recorder
F.linear = wrap(F.linear, "F.linear")
F.conv2d = wrap(F.conv2d, "F.conv2d")
for k in dir(ops):
    v = getattr(ops, k)
    if isinstance(v, types.FunctionType) and not k.startswith("_") and k not in ("linear", "split_linear_pays", "split_linear_supported", "split_mode",
                                                                                "linear_takes_split", "mlp_fused_ok", "linear_residual_fused"):
        setattr(ops, k, wrap(v, "ops." + k))
for fn in ("matmul", "bmm", "einsum", "softmax"):
    setattr(torch, fn, wrap(getattr(torch, fn), "torch." + fn))


def soak(mode, poison):
    a = A.complete(A.ARCHS["tiny3"])
    model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
    model.graph_replay = False
    if poison:
        with torch.no_grad():
            model.backbone.layers[1].blocks[0].norm1.bias[3] = 1.0e5
    g = torch.Generator().manual_seed(8)
    imgs = [torch.randint(0, 256, (3, 128, 192), generator=g, dtype=torch.uint8).cuda() for _ in range(3)]
    stop = threading.Event()

    def gemm_loop():
        st = torch.cuda.Stream()
        x = torch.randn(8192, 512, device="cuda")
        lin = torch.nn.Linear(512, 2048).cuda()
        with torch.cuda.stream(st), torch.no_grad():
            planes = _split_weight(lin.weight.detach(), "f16x3")
            while not stop.is_set():
                for _ in range(4):
                    _split_linear(x, planes, lin.bias, gelu=True)
                st.synchronize()

    bad_maps, nops = 0, 0
    with ops.split_mode(mode), torch.no_grad():
        refs = []
        logs = []
        for im in imgs:
            model.rba_scores([{"image": im}])
            state.update(mode="record", log=[], pos=0)
            refs.append(model.rba_scores([{"image": im}])[0].clone())
            logs.append(state["log"])
            state["mode"] = "off"
        nops = len(logs[0])
        th = threading.Thread(target=gemm_loop)
        th.start()
        try:
            for i in range(N):
                k = i % 3
                junk = [torch.full((n,), 1.0, device="cuda") for n in (1000 + 37 * (i % 11), 50000 + 1111 * (i % 7))]
                del junk
                state.update(mode="check", log=logs[k], pos=0)
                r = model.rba_scores([{"image": imgs[k]}])[0]
                state["mode"] = "off"
                bad_maps += int(not torch.equal(r, refs[k]))
        finally:
            stop.set()
            th.join()
    finite = all(bool(torch.isfinite(r).all()) for r in refs)
    rep = sorted(state["report"].values(), key=lambda e: e["op"])
    state["report"] = {}
    return {"arithmetic": mode, "poisoned_layer_norm_bias": poison, "forwards": N, "ops_per_forward": nops, "score_maps_finite": finite,
            "score_maps_that_differ_from_the_first": bad_maps, "non_reproducible_ops": rep[:12]}


if __name__ == "__main__":
    print(json.dumps(soak("bf16x6", True)), flush=True)
    print(json.dumps(soak("f16x3", False)), flush=True)
