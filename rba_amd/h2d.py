"""Host -> device hand-over of loader items.

A tensor that comes out of a `torch.utils.data.DataLoader` worker PROCESS (the reference's `DataLoader(dataset, batch_size=1,
num_workers=15)`, evaluate_ood.py:205-214, consumed by `x.to(device)` at support.py:371) lives in a shared-memory mapping this process has
not touched yet, and the first copy out of it is pathologically slow here (tools/shm_copy_probe.py, one 3 x 1024 x 2048 uint8 item):
`x.to(device)` 150-220 ms (the HIP runtime page-locks the mapping piece by piece for the DMA), a `copy_` fanned out over 128 intra-op threads
61 ms (they all take the first page faults of one mapping), a single-threaded memcpy 0.7-0.8 ms.  `to_device()` memcpys a pageable CPU tensor
into one of a few reusable page-locked staging buffers with ONE thread and issues the asynchronous copy from there; page-locked and device
tensors pass straight through.  One ring per (thread, device).  (The other half of the reference loop's 17-20 images/s -- GPU work submitted
while forked worker processes run -- is handled by `datasets.threaded`.)"""
import threading

import numpy as np
import torch

_tls = threading.local()


class _Stage:
    def __init__(self, depth=3):
        self.depth, self.slots, self.n = depth, [None] * depth, 0

    def put(self, x, device):
        nbytes = x.numel() * x.element_size()
        i = self.n % self.depth
        self.n += 1
        slot = self.slots[i]
        if slot is not None:
            slot[1].synchronize()                      # the copy that last read this buffer has finished
        if slot is None or slot[0].numel() < nbytes:
            slot = self.slots[i] = [torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True), torch.cuda.Event()]
        host = slot[0][:nbytes].view(x.dtype).view(x.shape)
        # ONE thread touches the source: torch's copy_ fans a 6 MB copy out over its intra-op threads, and 128 threads taking the first page
        # faults of one shared-memory mapping cost 60 ms where a plain memcpy costs 0.8 (tools/shm_copy_probe.py)
        np.copyto(slot[0][:nbytes].numpy(), x.view(-1).view(torch.uint8).numpy())
        out = torch.empty(x.shape, dtype=x.dtype, device=device)
        out.copy_(host, non_blocking=True)
        slot[1].record(torch.cuda.current_stream(device))
        return out


def to_device(x, device):
    """`x.to(device, non_blocking=True)` that never hands the HIP runtime pageable (in particular shared-memory) host pages."""
    device = torch.device(device)
    if not torch.is_tensor(x) or device.type != "cuda" or x.is_cuda or x.numel() == 0 or x.layout != torch.strided:
        return x.to(device, non_blocking=True) if torch.is_tensor(x) else x
    # is_shared() first: is_pinned() asks the HIP runtime about the pointer, and for a not-yet-touched shared-memory mapping that query alone
    # takes as long as the slow copy (tools/shm_copy_probe.py)
    if not x.is_shared() and x.is_pinned():
        return x.to(device, non_blocking=True)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    rings = getattr(_tls, "rings", None)
    if rings is None:
        rings = _tls.rings = {}
    ring = rings.get(device.index)
    if ring is None:
        ring = rings[device.index] = _Stage()
    return ring.put(x.contiguous(), device)
