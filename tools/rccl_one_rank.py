#!/usr/bin/env python3
"""Run this repository's collectives on RCCL in a ONE-rank `nccl` process group on device 0 (no gpurun box has two GPUs): all_gather of sizes + padded
all_gather_into_tensor on ragged DEVICE tensors, the histogram all_reduce, the pooled OoD metrics -- each equal to the single-process value.
Prints one line `RESULT {json}`.  tests/test_model_gpu.py::test_rccl_executes_the_metric_exchange_in_a_one_rank_group runs it."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from rba_amd import distributed as D
from rba_amd.metrics import ood_metrics

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), RBA_DIST_ONE_RANK_GROUP="1")
rank, world, local = D.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
g = torch.Generator(device="cuda").manual_seed(3)
out = {"backend": dist.get_backend(), "gathered": []}
for n in (0, 1, 1000, 123457):
    t = torch.randn(n, device="cuda", generator=g)
    assert D.all_gather_variable(t) is t                       # a 1-rank world returns early ...
    with D.force_collective():
        got = D.all_gather_variable(t)                         # ... unless told to run RCCL anyway
    assert got.is_cuda and (n == 0 or got.data_ptr() != t.data_ptr()) and torch.equal(got, t)
    out["gathered"].append(n)
s = torch.randn(200000, device="cuda", generator=g)
l = torch.rand(200000, device="cuda", generator=g) < 0.05
want = ood_metrics(s, l.to(torch.uint8))
with D.force_collective():
    pooled = D.pooled_ood_metrics(s, l)
    hist = D.histogram_ood_metrics(s, l)
assert pooled == want, (pooled, want)
assert all(abs(hist[k] - want[k]) < 2e-3 for k in want), (hist, want)
out["pooled"] = pooled
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
