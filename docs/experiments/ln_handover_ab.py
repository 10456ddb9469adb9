#!/usr/bin/env python3
"""The LayerNorm hand-over of Swin stage 3 (ops.linear_ln_out / linear_ln_in: the LayerNorm rides in the residual GEMM's epilogue and the next GEMM's epilogue)
against the launches it replaces (residual GEMM -> add_layer_norm(frag) -> GEMM), numerics against float64 and GPU time.  Usage: tools/ln_handover_ab.py [streams_hint]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops

hint = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops.set_concurrent_streams(hint)
busy = torch.randn(8192, 8192, device="cuda")


def timed(fn, reps=7, inner=20):
    ts = []
    for i in range(reps + 2):
        busy @ busy
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


torch.manual_seed(0)
for (M, C, Kp, Nn, gelu) in ((8192, 512, 512, 1536, False), (8192, 512, 2048, 2048, True), (8192, 768, 768, 2304, False), (8192, 768, 3072, 3072, True), (8200, 512, 512, 1536, False)):
    prod = torch.nn.Linear(Kp, C).cuda()                      # proj (Kp = C) or fc2 (Kp = 4 C)
    cons = torch.nn.Linear(C, Nn).cuda()                      # qkv or fc1
    g, b = (torch.rand(C, device="cuda") + 0.5), torch.randn(C, device="cuda") * 0.3
    a_in = ops.SplitActivations.pack(torch.randn(M, Kp, device="cuda"))
    r0 = torch.randn(M, C, device="cuda") * 2 + 0.3
    r0[:, 7] *= 40.0                                            # an outlier channel, as trained Swin residual streams have
    with torch.no_grad():
        r1 = ops.linear(a_in, prod, residual=r0.clone())
        y1 = ops.add_layer_norm(r1, g, b, 1e-5, frag=True)[1]
        q1 = ops.linear(y1, cons, gelu=gelu, split_out=gelu)
        r2, rows, stats = ops.linear_ln_out(a_in, prod, r0.clone())
        q2 = ops.linear_ln_in(rows, stats, (g, b, 1e-5), cons, gelu=gelu, split_out=gelu)
        ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(r1.double(), (C,), g.double(), b.double(), 1e-5), cons.weight.double(), cons.bias.double())
        if gelu:
            ref = torch.nn.functional.gelu(ref)
            q1, q2 = q1.unpack(), q2.unpack()
        t = r1.view(M, C // 128, 128).double()
        e_stats = max((stats[..., 0].double() - t.mean(-1)).abs().max().item(), ((stats[..., 1].double() - ((t - t.mean(-1, keepdim=True)) ** 2).sum(-1)).abs() / (1 + stats[..., 1].double())).max().item())
        line = (f"hint={hint} M={M} C={C} producer K={Kp} consumer N={Nn} gelu={gelu}: rows equal {torch.equal(r1, r2)}, image |d| {(rows.unpack() - r1).abs().max().item():.1e}, moments err {e_stats:.1e}, "
                f"max|err| vs float64: hand-over {(q2.double() - ref).abs().max().item():.2e} (rms {(q2.double() - ref).pow(2).mean().sqrt().item():.2e}), three launches {(q1.double() - ref).abs().max().item():.2e} "
                f"(rms {(q1.double() - ref).pow(2).mean().sqrt().item():.2e}), |ref| max {ref.abs().max().item():.1f}")
        rr = r0.clone()
        t_old = timed(lambda: ops.linear(ops.add_layer_norm(ops.linear(a_in, prod, residual=rr), g, b, 1e-5, frag=True)[1], cons, gelu=gelu, split_out=gelu))
        t_new = timed(lambda: ops.linear_ln_in(*ops.linear_ln_out(a_in, prod, rr)[1:], (g, b, 1e-5), cons, gelu=gelu, split_out=gelu))
        t_p_old, t_p_new = timed(lambda: ops.linear(a_in, prod, residual=rr)), timed(lambda: ops.linear_ln_out(a_in, prod, rr))
    print(line + f" | time: three launches {t_old:6.1f} us, hand-over {t_new:6.1f} us (producer alone {t_p_old:5.1f} -> {t_p_new:5.1f})", flush=True)
