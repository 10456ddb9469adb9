#!/usr/bin/env python3
"""Does torch's process-based DataLoader scale on this box with the drop-in dataset classes, and does an initialised GPU context in the parent change it?
python tools/dataloader_probe.py [n_images]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.evaluator_bench as EB
from rba_amd.datasets import get_dataset, ThreadLoader
from torch.utils.data import DataLoader

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    work = "/tmp/rba_dl_probe"
    EB.make_dataset(work + "/data", n)
    ds = get_dataset("fishyscapes_laf", work + "/data")
    print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())


    def run(loader, tag):
        t = time.perf_counter(); k = 0; t_first = None
        for b in loader:
            k += 1
            if t_first is None: t_first = time.perf_counter() - t
        dt = time.perf_counter() - t
        print(f"{tag:40s} {k} items  {k / dt:6.1f} items/s  (first item after {t_first:.2f} s)")


    run(DataLoader(ds, batch_size=1, num_workers=0), "workers 0")
    run(DataLoader(ds, batch_size=1, num_workers=15), "workers 15, no GPU context yet")
    run(ThreadLoader(ds, 8), "ThreadLoader(8)")
    if torch.cuda.is_available():
        x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
        run(DataLoader(ds, batch_size=1, num_workers=15), "workers 15, GPU context in the parent")
        run(DataLoader(ds, batch_size=1, num_workers=15, persistent_workers=False, prefetch_factor=4), "workers 15, prefetch 4")
        run(ThreadLoader(ds, 8), "ThreadLoader(8), GPU context")


if __name__ == "__main__":
    main()
