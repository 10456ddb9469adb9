#!/usr/bin/env python3
"""Which op of the forward is not reproducible from run to run?  Records every library call (F.linear, F.conv2d, torch.matmul ...)
and every rba_amd.ops launch of one forward with its inputs and outputs, then replays the forward (after perturbing the caching
allocator, and optionally after running other shapes / models in between) and reports each op whose inputs are bitwise equal to the
recorded ones but whose output is not.

  python tools/op_replay.py [arch] [H W] [--disturb]

Found with it: MIOpen's 1x1 convolution on a ONE-pixel map (res5 of a 32x32 image) differs by 2e-7 from run to run; at every real
image size all ops are bitwise reproducible."""
import os
import sys
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from rba_amd import arch as A, ops
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

argv = [v for v in sys.argv[1:] if not v.startswith("--")]
disturb = "--disturb" in sys.argv
name = argv[0] if argv else "tiny1"
h, w = (int(argv[1]), int(argv[2])) if len(argv) > 2 else (16, 32)


def build(name):
    a = A.complete(A.ARCHS[name])
    return load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()


m = build(name)
g = torch.Generator().manual_seed(3)
img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8).cuda()
log, mode, pos, report, active = [], "record", 0, [], True


def tens(args):
    out = []
    for v in args:
        if isinstance(v, torch.Tensor):
            out.append(v)
        elif isinstance(v, (list, tuple)):
            out += tens(v)
        elif isinstance(v, torch.nn.Module):
            out += list(v.parameters())
    return out


def wrap(fn, label):
    def inner(*args, **kw):
        global pos
        r = fn(*args, **kw)
        if not active:
            return r
        ins = [t.detach().clone() for t in tens(args) + tens(list(kw.values()))]
        outs = [t.detach().clone() for t in tens([r])]
        if mode == "record":
            log.append((label, ins, outs))
        else:
            lab, i0, o0 = log[pos]
            assert lab == label, (lab, label)
            same_in = len(i0) == len(ins) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(i0, ins))
            same_out = all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(o0, outs))
            if same_in and not same_out:
                d = max(float((x.float() - y.float()).abs().max()) for x, y in zip(o0, outs))
                report.append((pos, label, [tuple(t.shape) for t in ins][:4], d))
            pos += 1
        return r
    return inner


F.linear = wrap(F.linear, "F.linear")
F.conv2d = wrap(F.conv2d, "F.conv2d")
for k in dir(ops):
    v = getattr(ops, k)
    if isinstance(v, types.FunctionType) and not k.startswith("_") and k not in ("linear", "split_linear_pays", "split_linear_supported"):
        setattr(ops, k, wrap(v, "ops." + k))
for fn in ("matmul", "bmm", "einsum", "softmax"):
    setattr(torch, fn, wrap(getattr(torch, fn), "torch." + fn))


def run(model=m, image=img):
    with torch.no_grad():
        r = model.rba_scores([{"image": image}])[0]
    torch.cuda.synchronize()
    return r


run()
log.clear()
base = run().clone()
mode = "check"
others = []
if disturb:
    active = False
    others = [(build("tiny1"), (60, 90)), (build("tiny3"), (40, 72)), (m, (h // 2, w // 2)), (m, (h + 64, w + 32))]
    active = True
ndiff = 0
for i in range(6):
    junk = [torch.full((n,), 1.0, device="cuda") for n in (1000 + 37 * i, 50000 + 1111 * i, 2_000_000 + 999 * i)]
    del junk
    if disturb:
        active = False
        for mod, (hh, ww) in others:
            run(mod, torch.randint(0, 256, (3, hh, ww), dtype=torch.uint8).cuda())
        active = True
    pos = 0
    ndiff += int(not torch.equal(run(), base))
seen = set()
for p, label, shapes, d in report:
    if (p, label) not in seen:
        seen.add((p, label))
        print("op", p, label, shapes, "max |d|", d)
print(f"{name} {h}x{w}: ops per forward {len(log)}, non-reproducible op instances {len(seen)}, replays with a different score map {ndiff}/6")
