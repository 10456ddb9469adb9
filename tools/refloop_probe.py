#!/usr/bin/env python3
"""Where does the reference's batch-1 scoring loop spend its time per image with a 15-process DataLoader?  (tools/evaluator_bench.py: 20 images/s with it,
83 with ThreadLoader(8), while the DataLoader alone delivers 60-75 items/s: tools/dataloader_probe.py.)   python tools/refloop_probe.py [n_images]"""
import os, sys, time, threading, queue
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import yaml
    import tools.evaluator_bench as EB
    from rba_amd import arch as A
    from rba_amd import evaluate_ood as E
    from rba_amd.datasets import get_dataset, ThreadLoader
    from rba_amd.h2d import to_device
    from rba_amd.support import OODEvaluator
    from torch.utils.data import DataLoader
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    work = "/tmp/rba_refloop_probe"
    EB.make_dataset(os.path.join(work, "data"), n)
    mdir = os.path.join(work, "ckpts", "swin_b_1dl")
    os.makedirs(mdir, exist_ok=True)
    cfg = {"MODEL": {"META_ARCHITECTURE": "MaskFormer", "BACKBONE": {"NAME": "D2SwinTransformer"},
                     "SWIN": {"EMBED_DIM": 128, "DEPTHS": [2, 2, 18, 2], "NUM_HEADS": [4, 8, 16, 32], "WINDOW_SIZE": 12},
                     "SEM_SEG_HEAD": {"DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res5"]}, "MASK_FORMER": {"DEC_LAYERS": 2}}}
    with open(os.path.join(mdir, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    a = A.complete(A.ARCHS["swin_b_1dl"])
    torch.save({"model": A.seeded_weights(a, 0)}, os.path.join(mdir, "model_final.pth"))
    ds = get_dataset("fishyscapes_laf", os.path.join(work, "data"))
    model = E.get_model(os.path.join(mdir, "config.yaml"), os.path.join(mdir, "model_final.pth"))
    dev = torch.device("cuda")
    for i in range(3):
        model.rba_scores([{"image": ds[i][0].to(dev)}])
    torch.cuda.synchronize()

    def loop(loader, tag, label_to_numpy=True, to_dev=True, score=True, staged=False):
        it = iter(loader)
        tn = th = tl = ts = 0.0
        k = 0
        t00 = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            try:
                x, y = next(it)
            except StopIteration:
                break
            t1 = time.perf_counter()
            if to_dev:
                x = to_device(x, dev) if staged else x.to(dev, non_blocking=True)
            t2 = time.perf_counter()
            if label_to_numpy:
                g = np.asarray(y.cpu())
            t3 = time.perf_counter()
            if score:
                s = model.rba_scores([{"image": x[0]}])[0]
            t4 = time.perf_counter()
            tn += t1 - t0; th += t2 - t1; tl += t3 - t2; ts += t4 - t3; k += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t00
        print(f"{tag:44s} {k / dt:6.1f} images/s   per image: next() {tn / k * 1e3:6.1f} ms  .to(device) {th / k * 1e3:5.1f}  label {tl / k * 1e3:5.1f}  rba_scores() {ts / k * 1e3:5.1f}", flush=True)

    class Prefetch:
        """iterate any loader from a helper thread, hand items over through a bounded queue"""
        def __init__(self, loader, depth=4):
            self.loader, self.depth = loader, depth
        def __iter__(self):
            q = queue.Queue(self.depth)
            def work():
                for item in self.loader:
                    q.put(item)
                q.put(None)
            threading.Thread(target=work, daemon=True).start()
            while True:
                item = q.get()
                if item is None:
                    return
                yield item

    mk = lambda **kw: DataLoader(ds, shuffle=False, batch_size=1, num_workers=15, timeout=300, **kw)
    loop(mk(), "DataLoader(15): iterate only", label_to_numpy=False, to_dev=False, score=False)
    loop(mk(), "DataLoader(15): + .to(device)", label_to_numpy=False, score=False)
    loop(mk(), "DataLoader(15): + label -> numpy", score=False)
    loop(mk(), "DataLoader(15): full loop")
    loop(Prefetch(mk()), "DataLoader(15) behind a prefetch thread")
    loop(mk(pin_memory=True), "DataLoader(15, pin_memory=True)")
    loop(ThreadLoader(ds, 8), "ThreadLoader(8): full loop")
    loop(mk(), "DataLoader(15): rba_amd.h2d.to_device", staged=True)
    # the staged hand-over taken apart
    pins = [torch.empty(3 * 1024 * 2048, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
    evs = [None] * 3
    acc = {}
    k = 0
    for x, y in mk():
        t = [time.perf_counter()]
        i = k % 3
        if evs[i] is not None:
            evs[i].synchronize()
        t.append(time.perf_counter())
        src = x.view(-1).numpy()
        t.append(time.perf_counter())
        np.copyto(pins[i].numpy(), src)
        t.append(time.perf_counter())
        out = torch.empty(x.shape, dtype=x.dtype, device=dev)
        t.append(time.perf_counter())
        out.copy_(pins[i].view(x.shape), non_blocking=True)
        t.append(time.perf_counter())
        evs[i] = torch.cuda.Event(); evs[i].record()
        t.append(time.perf_counter())
        s_ = model.rba_scores([{"image": out[0]}])[0]
        t.append(time.perf_counter())
        for name, a, b in zip(("event wait", "numpy view", "memcpy shm -> pinned", "device alloc", "async H2D", "event record", "rba_scores"), t[:-1], t[1:]):
            acc[name] = acc.get(name, 0.0) + b - a
        k += 1
    print("staged hand-over per image: " + "  ".join(f"{n_} {v / k * 1e3:.2f} ms" for n_, v in acc.items()), flush=True)
    ev = OODEvaluator(model, E.get_logits, E.get_RbA)
    for tag, ld in (("OODEvaluator loop, DataLoader(15)", mk()), ("OODEvaluator loop, ThreadLoader(8)", ThreadLoader(ds, 8))):
        t0 = time.perf_counter()
        sc, gt = ev.compute_anomaly_scores(ld, device=dev, upper_limit=n)
        dt = time.perf_counter() - t0
        print(f"{tag:44s} {len(sc) / dt:6.1f} images/s", flush=True)


if __name__ == "__main__":
    main()
