#!/usr/bin/env python3
"""Second probe of the GroupNorm-fold mystery (profiles/r05_gnfold_select.txt): the probe builds of csrc/tune/gnf_form*_dbg*.hip -- form 0 (product: max + poison)
and form 1 (max(y, f) + (y - y), the one that goes wrong), each with and without a dump of what every thread staged: raw rows xr, the values it stored xst, and its
coefficients a, b, per (tile, k block, thread).  The host then says WHICH of them is wrong where the output is wrong.   python tools/gnfold_probe2.py [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import _tune
from rba_amd import ops

lib = _tune.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator().manual_seed(1 + 131072 + 256 + 256)
B, P, K, N, G = 1, 131072, 256, 256, 32
x = torch.randn(B, P, K, generator=g) * 3 + 0.7
w, b = torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
ga, be = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
xd = x.cuda()
p3 = ops.split_weight(w.cuda(), mode="f16x3")
gac, bec, bc = ga.cuda(), be.cuda(), b.cuda()
relu = 0
yref = F.group_norm(xd.double().permute(0, 2, 1), G, gac.double(), bec.double(), 1e-5).permute(0, 2, 1)      # [B, P, K]
ref = (yref.reshape(B * P, K) @ w.cuda().double().t() + bc.double()).view(B, P, N).permute(0, 2, 1)
mr = ops.group_norm_nhwc_stats(xd, G, 1e-5)
M, NT, NB = B * P, N // 128, K // 32
MT = M // 128
st = torch.cuda.current_stream().cuda_stream


def run(form, dbg_on, suffix=""):
    fn = getattr(lib, f"rba_gnf_probe_form{form}_dbg{dbg_on}{suffix}")
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
    fn.restype = ctypes.c_int
    out = torch.empty(B, N, P, device="cuda")
    dbg = torch.zeros(MT * NT * NB * 256 * 40, device="cuda") if dbg_on else None
    rc = fn(xd.data_ptr(), mr.data_ptr(), gac.data_ptr(), bec.data_ptr(), G, relu, p3.data_ptr(), bc.data_ptr(), out.data_ptr(), M, N, K, P,
            dbg.data_ptr() if dbg_on else 0, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out, dbg


for form, dbg_on, suffix in ((0, 0, ""), (0, 1, ""), (1, 0, ""), (1, 1, ""), (0, 1, "_unpacked"), (1, 0, "_unpacked")):
    if True:
        for rep in range(reps):
            out, dbg = run(form, dbg_on, suffix)
            e = (out.double() - ref).abs().amax(dim=(0, 1))                 # per pixel
            bad = torch.nonzero(e > 1e-3).flatten()
            line = f"form {form} dump {dbg_on}{suffix} run {rep}: max err {e.max().item():.3g}, bad pixels {bad.numel()}"
            if dbg_on:
                d = dbg.view(MT, NT, NB, 256, 40)
                xr, xs, a, bb = d[..., :16].view(MT, NT, NB, 256, 4, 4), d[..., 16:32].view(MT, NT, NB, 256, 4, 4), d[..., 32:36], d[..., 36:40]
                # what the thread should have had: row (tid >> 3) + 32 q of tile mt, channels 32 blk + 4 (tid & 7) .. + 3
                tid = torch.arange(256, device="cuda")
                rows = (tid[:, None] >> 3) + 32 * torch.arange(4, device="cuda")[None, :]                            # [256, 4]
                ch = 4 * (tid & 7)                                                                                    # [256]
                xt = xd.view(MT, 128, NB, 8, 4)                                                                      # [mt, row, blk, chunk, 4]
                want_x = xt[:, rows, :, (tid & 7)[:, None], :]                                                       # advanced indices split by a slice: [256, 4, MT, NB, 4]
                want_x = want_x.permute(2, 3, 0, 1, 4)[:, None].expand(MT, NT, NB, 256, 4, 4)
                yt = yref.float().view(MT, 128, NB, 8, 4)
                want_y = yt[:, rows, :, (tid & 7)[:, None], :].permute(2, 3, 0, 1, 4)[:, None].expand(MT, NT, NB, 256, 4, 4)
                recomputed = torch.addcmul(bb[..., None, :], xr, a[..., None, :])                                    # x a + b from the thread's own dumped inputs (fp32, not fused: ~1e-6)
                bad_x = (xr != want_x)
                bad_s = (xs - want_y).abs() > 1e-3
                bad_r = (recomputed - want_y).abs() > 1e-3
                incons = (xs - recomputed).abs() > 1e-3                                                              # stored value disagrees with the thread's own inputs
                line += (f" | dumped: raw rows wrong {int(bad_x.sum())}, stored values wrong {int(bad_s.sum())}, x a + b from the dumped inputs wrong {int(bad_r.sum())}, "
                         f"stored != own inputs {int(incons.sum())}")
                # expected coefficients of (block, thread): channels 32 blk + 4 (tid & 7) + i, group = channel // cpg
                chn = (32 * torch.arange(NB, device="cuda")[:, None, None] + 4 * (tid & 7)[None, :, None] + torch.arange(4, device="cuda")[None, None, :])   # [NB, 256, 4]
                mean, rstd = mr.view(G, 2)[:, 0], mr.view(G, 2)[:, 1]
                a_exp = gac[chn] * rstd[chn // (K // G)]
                b_exp = torch.addcmul(bec[chn], -mean[chn // (K // G)], a_exp)
                bad_a = (a - a_exp[None, None]).abs() > 1e-5
                bad_b = (bb - b_exp[None, None]).abs() > 1e-4
                line += f" | coefficients: a wrong {int(bad_a.sum())}, b wrong {int(bad_b.sum())}"
                for (mt_, nt_, blk_, tid_, i_) in torch.nonzero(bad_a | bad_b)[:5].tolist():
                    av, bv = a[mt_, nt_, blk_, tid_, i_].item(), bb[mt_, nt_, blk_, tid_, i_].item()
                    # does the wrong value belong to another block / thread?
                    hit_a = torch.nonzero((a_exp - av).abs() < 1e-7)[:3].tolist()
                    hit_b = torch.nonzero((b_exp - bv).abs() < 1e-7)[:3].tolist()
                    line += (f"\n      tile ({mt_},{nt_}) blk {blk_} tid {tid_} i {i_}: a {av:.6f} (want {a_exp[blk_, tid_, i_].item():.6f}) b {bv:.6f} (want {b_exp[blk_, tid_, i_].item():.6f})"
                             f"  a matches (blk, tid, i) {hit_a}  b matches {hit_b}")
                if int(bad_s.sum()):
                    idx = torch.nonzero(bad_s)[:6].tolist()
                    line += f" first (mt, nt, blk, tid, q, i): {idx}"
                    lanes = torch.nonzero(bad_s)[:, 3] & 63
                    line += f" lanes {sorted(set(lanes.tolist()))[:20]}"
            print(line, flush=True)
