#!/usr/bin/env python3
"""K7 (rba_swin_attn_block_f32: norm1 -> qkv -> window attention -> proj -> + shortcut -> norm2 in one kernel) against the unfused sequence of
launches it replaces (add_layer_norm -> K6 qkv -> K5 -> K6 proj + residual -> add_layer_norm) on the Swin-B stage shapes it has kernels for.
Usage: tools/k7_ab.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd import ops
from rba_amd.modeling.backbone.swin import SwinTransformerBlock

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
busy = torch.randn(8192, 8192, device="cuda")
ws = 12


def timed(fn, reps=7, inner=10):
    ts = []
    for i in range(reps + 2):
        busy @ busy                              # launches queue behind a long kernel: events bracket GPU time, not host latency
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


for (H, W, nH) in ((256, 512, 4), (128, 256, 8), (64, 128, 16)):
    C = nH * 32
    if not ops.swin_attn_block_ok(C, nH, ws):
        if ops.swin_attn_qkv_ok(C, nH, ws):                     # attention-only form (no proj): against add_layer_norm -> K6 qkv -> K5
            torch.manual_seed(0)
            for shift in (0, 6):
                blk = SwinTransformerBlock(C, nH, ws, shift, 4.0).cuda().eval()
                x = torch.randn(B, H * W, C, device="cuda")
                a = blk.attn
                bias, frag = a.gathered_bias()
                img = a.block_image()
                n1 = (blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
                M = B * H * W

                def fq():
                    return ops.swin_attn_qkv(x, n1, img, a.qkv.bias, frag, H, W, ws, shift)

                def uq():
                    y = ops.add_layer_norm(x, *n1, frag=ops.linear_takes_split(M, 3 * C, C))[1]
                    return ops.swin_window_attn(ops.linear(y, a.qkv), a.qkv.bias, bias, H, W, nH, ws, shift, bias_frag=frag, split_out=True)
                with torch.no_grad():
                    e = (fq().unpack() - uq().unpack()).abs().max().item()
                    tf, tu = timed(fq), timed(uq)
                print(f"B={B} {H}x{W} C={C} shift={shift}: attention-only fused {tf:7.1f} us   LN + qkv + K5 {tu:7.1f} us   ({tu / tf:.2f}x)   max diff {e:.2e}", flush=True)
        continue
    torch.manual_seed(0)
    for shift in (0, 6):
        blk = SwinTransformerBlock(C, nH, ws, shift, 4.0).cuda().eval()
        with torch.no_grad():
            for p in blk.parameters():
                if p.dim() == 1:
                    p.normal_(0, 0.2)
            blk.norm1.weight.add_(1.0)
            blk.norm2.weight.add_(1.0)
        x0 = torch.randn(B, H * W, C, device="cuda")
        a = blk.attn
        bias, frag = a.gathered_bias()
        img = a.block_image()
        n1 = (blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        n2 = (blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        M = B * H * W

        def fused():
            return ops.swin_attn_block(x, n1, img, a.qkv.bias, frag, a.proj.bias, H, W, ws, shift, norm2=n2)

        def fused_no_y2():
            return ops.swin_attn_block(x, n1, img, a.qkv.bias, frag, a.proj.bias, H, W, ws, shift)

        def unfused():
            xx, y = ops.add_layer_norm(x, *n1, frag=ops.linear_takes_split(M, 3 * C, C))
            qkv = ops.linear(y, a.qkv)
            y = ops.swin_window_attn(qkv, a.qkv.bias, bias, H, W, nH, ws, shift, bias_frag=frag, split_out=ops.linear_takes_split(M, C, C))
            xx = ops.linear(y, a.proj, residual=x)
            return xx, ops.add_layer_norm(xx, *n2)[1]

        with torch.no_grad():
            x = x0.clone()
            xf, yf = fused()
            xf, yf = xf.clone(), yf.clone()
            x = x0.clone()
            xu, yu = unfused()
            err = (xf - xu).abs().max().item(), (yf - yu).abs().max().item()
            x = x0.clone()
            tf = timed(fused)
            x = x0.clone()
            tn = timed(fused_no_y2)
            x = x0.clone()
            tu = timed(unfused)
        print(f"B={B} {H}x{W} C={C} shift={shift}: fused {tf:7.1f} us (without norm2 {tn:7.1f})   unfused {tu:7.1f} us   ({tu / tf:.2f}x)   max|fused - unfused| x {err[0]:.2e} y2 {err[1]:.2e}",
              flush=True)
