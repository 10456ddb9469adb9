"""Throughput of MaskFormer.predict for batches of 1, 2, 4 images in ONE forward (one stream):  python tools/batch_probe.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rba_amd import arch as A, ops
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer
torch.manual_seed(0)
a = A.complete(A.ARCHS["swin_b_1dl"])
m = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).to("cuda").eval()
imgs = [torch.randint(0, 256, (3, 1024, 2048), dtype=torch.uint8, device="cuda") for _ in range(4)]
with torch.no_grad():
    for B in (1, 2, 4):
        batch = [{"image": imgs[i]} for i in range(B)]
        for _ in range(2):
            out = m.predict(batch)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 6
        for _ in range(n):
            out = m.predict(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print(f"B={B}: {dt*1e3:.2f} ms per forward, {B/dt:.1f} images/s (predict only)")
