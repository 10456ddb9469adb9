"""Where a sample's time goes in the evaluator's process loader (rba_amd.datasets.ProcessDecoder), without a model: decode in the child, the
write to /dev/shm, the read into (page-locked) memory in the parent; beside the thread loader on the same files.
  python tools/decode_procs_probe.py [images] [workdir]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from evaluator_bench import make_dataset
from rba_amd import datasets as DS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/rba_decode_probe"
make_dataset(os.path.join(work, "data"), n)
ds = DS.get_dataset("fishyscapes_laf", os.path.join(work, "data"))
pin = torch.cuda.is_available()
if pin:
    torch.zeros(1, device="cuda")
out = {"images": n, "pinned": pin, "shm": os.popen("df -h /dev/shm | tail -1").read().split()}
for procs in (4, 8, 16):
    with DS.ProcessDecoder(procs) as pd:
        time.sleep(1.0)                                              # children imported
        t = time.perf_counter()
        for _ in pd.items(ds, range(n), pin=pin):
            pass
        dt = time.perf_counter() - t
        k = max(1, pd.stats["items"])
        out[f"processes_{procs}"] = {"images_per_s": round(n / dt, 1), "per_sample_ms": {a: round(v / k * 1e3, 2) for a, v in pd.stats.items() if a != "items"}}
for th in (8,):
    t = time.perf_counter()
    for _ in DS.prefetch(ds, range(n), th, pin=pin, raw=True):
        pass
    out[f"threads_{th}"] = {"images_per_s": round(n / (time.perf_counter() - t), 1)}
print(json.dumps(out))
