"""Soak test of the evaluator's default scorer, the fused x4-upsample + RbA reduction with the class contraction on the matrix pipe
(rba_reduce_up4_f32, score only -> rba_reduce_up4_mx_kernel): VERDICT r3 "weak #1".

  kernel leg: `launches` launches at BASELINE C2's map (100 x 256 x 512 -> 1024 x 2048) and at C5's (100 x 180 x 320 -> 720 x 1280), issued
              round-robin from three HIP streams while a fourth stream runs a K6 GEMM loop (the matrix pipe and the vector L1 busy with
              someone else's waves); every output must be bit-equal to launch 0 of its map.
  model leg:  `forwards` calls of MaskFormer.rba_scores (Swin-B 1dl, graph replay on) per stream, alternating over three streams from one issuing thread
              (as the evaluator's scoring loop does); every score map bit-equal to the first one.

On a mismatch: which launch, how many pixels, their (row, column) bounding box, whether they form whole 128-pixel wave tiles, max |d|.
`python tools/k1_soak.py [launches] [forwards]` prints one JSON line (RBA_HIP_LIB selects the build: A/B against tools/ab/librba_hip_r3.so);
`soak_kernel` / `soak_model` are what tests/test_kernels_gpu.py and tests/test_model_gpu.py call with smaller counts."""
import json
import os
import sys
import threading

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def describe(out, ref):
    bad = (out != ref)
    idx = bad.nonzero()
    d = (out.double() - ref.double()).abs()
    rows, cols = idx[:, 0], idx[:, 1]
    # a wave of rba_reduce_up4_mx_kernel owns 128 consecutive pixels of one row: are the bad pixels whole tiles / lane halves?
    tiles = torch.unique(rows * 100000 + cols // 128)
    return {"pixels": int(bad.sum()), "rows": [int(rows.min()), int(rows.max())], "cols": [int(cols.min()), int(cols.max())],
            "wave_tiles_touched": int(tiles.numel()), "max_abs_diff": float(d.max()), "nan": int(torch.isnan(out).sum()),
            "first": [[int(r), int(c)] for r, c in idx[:8].tolist()]}


def soak_kernel(launches=2000, maps=((256, 512, 1024, 2048), (180, 320, 720, 1280)), gemm=True, chunk=48):
    from rba_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    cases = []
    for h, w, ch, cw in maps:
        low = (torch.randn(100, h, w, generator=g) * 5).to(dev)
        prob = F.softmax(torch.randn(100, 20, generator=g) * 3, -1)[:, :-1].contiguous().to(dev)
        ref = ops.rba_reduce_up4(low, prob, (ch, cw))[0]
        cases.append((low, prob, (ch, cw), ref))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    side = torch.cuda.Stream()
    stop = threading.Event()

    def gemm_loop():                                                    # stage-3 fc1-shaped K6 launches until told to stop
        x = torch.randn(8192, 512, device=dev)
        lin = torch.nn.Linear(512, 2048).to(dev)
        with torch.cuda.stream(side), torch.no_grad():
            while not stop.is_set():
                for _ in range(8):
                    ops.linear(x, lin)
                side.synchronize()

    th = threading.Thread(target=gemm_loop) if gemm else None
    if th:
        th.start()
    mismatches, done = [], 0
    try:
        while done < launches and not mismatches:
            n = min(chunk, launches - done)
            outs = []
            for i in range(n):
                ci = (done + i) % len(cases)
                low, prob, crop, ref = cases[ci]
                with torch.cuda.stream(streams[(done + i) % 3]):
                    outs.append((done + i, ci, ops.rba_reduce_up4(low, prob, crop)[0]))
            for s in streams:
                s.synchronize()
            for li, ci, o in outs:
                if not torch.equal(o, cases[ci][3]):
                    mismatches.append({"launch": li, "map": list(maps[ci]), **describe(o, cases[ci][3])})
            done += n
    finally:
        stop.set()
        if th:
            th.join()
    return {"launches": done, "streams": 3, "gemm_beside": bool(gemm), "mismatching_launches": len(mismatches), "mismatches": mismatches[:4]}


def soak_model(forwards=200, arch="swin_b_1dl", hw=(1024, 2048), nstreams=3):
    from rba_amd import arch as A
    from rba_amd.checkpoint import load_checkpoint
    from rba_amd.maskformer_model import MaskFormer
    a = A.complete(A.ARCHS[arch])
    model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
    model.graph_replay = True
    g = torch.Generator().manual_seed(5)
    image = torch.randint(0, 256, (3,) + tuple(hw), generator=g, dtype=torch.uint8).cuda()
    ref = model.rba_scores([{"image": image}])[0].clone()
    torch.cuda.synchronize()
    # one issuing thread, the forwards alternating over the streams (what rba_amd.evaluate_ood's scoring loop does): rounds 0-1 run eagerly and
    # capture one graph per stream, later rounds replay the three graphs concurrently
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    bad, pend = [], []
    for i in range(forwards):
        for k, st in enumerate(streams):
            if i == 0:
                st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                pend.append((i, k, model.rba_scores([{"image": image}])[0]))
        if len(pend) >= 24 or i == forwards - 1:
            for st in streams:
                st.synchronize()
            for fi, k, r in pend:
                if not torch.equal(r, ref):
                    bad.append({"forward": fi, "stream": k, **describe(r, ref)})
            pend = []
            if len(bad) >= 4:
                break
    return {"forwards_per_stream": forwards, "streams": nstreams, "graph_replay": True, "live_graphs": model.live_graphs(),
            "mismatching_forwards": len(bad), "mismatches": bad[:4]}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    out = {"lib": os.environ.get("RBA_HIP_LIB", "rba_amd/csrc/librba_hip.so"), "kernel": soak_kernel(n)}
    if m:
        out["model"] = soak_model(m)
    print(json.dumps(out))
