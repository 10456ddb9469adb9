"""images/s of the whole scoring path (MaskFormer.rba_scores: forward + fused x4 upsample + K1) for S HIP streams x B images per forward, each
stream's forward replayed from a hipGraph:  python tools/batch_stream_probe.py [arch] [H W]   (COMBOS=3x1,2x3 selects stream x batch pairs)
Round-3 result (profiles/r03_batch_streams.txt): 3 x 1: 134.1, 3 x 2: 135.1, 2 x 3: 137.8, 3 x 3: 138.7, 2 x 4: 138.8 images/s -- batching buys 1-3 % over
three batch-1 streams, so bench.py and the evaluator stay batch-1 like the reference's loop.  Known problem of THIS TOOL: capturing a second set of graphs
for batch 4 in one process (COMBOS=1x4,2x4) ends in a GPU memory access fault, with every round-3 kernel switched off too (TOGGLE=...), not in eager mode
and not when 2x4 is the first configuration: unresolved, outside what the product does (one graph per image shape and stream, batch 1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _knobs  # noqa: F401  (knob-writing tool: run on librba_hip_knobs.so)
from rba_amd import arch as A
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer

arch = sys.argv[1] if len(sys.argv) > 1 else "swin_b_1dl"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 2048)
a = A.complete(A.ARCHS[arch])
m = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).to("cuda").eval()
m.graph_replay = False
import ctypes
from rba_amd import _lib, ops
tg = os.environ.get("TOGGLE", "")
if "noh3q" in tg:
    ctypes.c_int.in_dll(_lib.load(), "rba_k6_variant").value = 1
if "nomsda" in tg:
    ops.msda_fused_ok = lambda D, L, P, S=None, M=None: False
if "nofront" in tg:
    m.fused_front_end = False
if "noup4" in tg:
    m.fused_upsample = False
if "nosparse" in tg:
    m.sem_seg_head.predictor.sparse_intermediate_heads = False
if "nomlp" in tg:
    ops.MLP_FUSED_MIN_ROWS = 1 << 40
g = torch.Generator().manual_seed(0)
imgs = [torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g).cuda() for _ in range(12)]
keep = []
ALL_STREAMS = [torch.cuda.Stream() for _ in range(4)]          # created once: see the note at the end of this file
with torch.no_grad():
    combos = [tuple(int(v) for v in c.split("x")) for c in os.environ["COMBOS"].split(",")] if os.environ.get("COMBOS") else ((1, 1), (3, 1), (1, 2), (2, 2), (3, 2), (1, 3), (2, 3), (1, 4), (2, 4), (3, 4))
    for S, B in combos:
        streams = ALL_STREAMS[:S]
        graphs = []
        for si, st in enumerate(streams):
            batch = [{"image": imgs[(si * B + i) % len(imgs)].clone()} for i in range(B)]
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                m.rba_scores(batch)
                m.rba_scores(batch)
            torch.cuda.synchronize()
            if os.environ.get("EAGER"):
                graphs.append((None, st, batch))
                continue
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st, capture_error_mode="thread_local"):
                out = m.rba_scores(batch)
            graphs.append((gr, st, out))
        torch.cuda.synchronize()
        n = 8
        for rep in range(2):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                for gr, st, out in graphs:
                    with torch.cuda.stream(st):
                        if gr is None:
                            m.rba_scores(out)
                        else:
                            gr.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        print(f"streams {S} x batch {B}: {n * S * B / dt:7.1f} images/s", flush=True)
        keep.append(graphs) if os.environ.get("KEEP") else None
        del graphs
        torch.cuda.empty_cache()
