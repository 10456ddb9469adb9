"""Multi-GPU evaluation: images shard across ranks (one process per GPU, no communication in the forward), and the
OoD metrics -- which rank ALL pixels of ALL images together (support.py:275-290), so they are not averages of
per-shard values -- are pooled with one RCCL all-gather of (score, label) pairs over xGMI.  New functionality
w.r.t. the reference, whose evaluation is single process.  Works on gloo (CPU tests) and nccl=RCCL.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RBA_DIST_ONE_RANK_GROUP=1: initialise the process group even for ONE rank, so that a 1-GPU box can run the collective code on RCCL itself
    # (with force_collective() below); a normal single-process run never touches torch.distributed
    if (world > 1 or os.environ.get("RBA_DIST_ONE_RANK_GROUP") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            if local >= torch.cuda.device_count():
                raise RuntimeError(f"LOCAL_RANK {local} but only {torch.cuda.device_count()} visible GPU(s)")
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist format) -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(index, sysfs="/sys/bus/pci/devices"):
    """CPUs of the NUMA node the HIP device `index` hangs off: /sys/bus/pci/devices/<domain:bus:device.0>/local_cpulist of the device's PCI address
    (torch.cuda.get_device_properties: pci_domain_id / pci_bus_id / pci_device_id).  None when the address or the sysfs entry is not there."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(os.path.join(sysfs, bdf, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except Exception:                                        # noqa: BLE001 -- placement is an optimisation
        return None


def plan_rank_cpus(local_rank, local_world, local_cpus_of, allowed):
    """CPU set for the process of `local_rank`: the CPUs local to its GPU (local_cpus_of(rank) -> list | None) that this process may use (`allowed`), divided
    evenly among the local ranks whose GPUs report the SAME set (8 GPUs on 2 sockets: four ranks share a socket, each gets a quarter of it, SMT siblings
    included as the kernel numbers them).  Falls back to an even split of `allowed` by local rank.  -> (sorted cpu list, source string)"""
    allowed = sorted(allowed)
    mine = local_cpus_of(local_rank)
    if mine:
        mine = [c for c in mine if c in set(allowed)]
    if mine:
        peers = [r for r in range(local_world) if (local_cpus_of(r) or None) is not None and sorted(local_cpus_of(r)) == sorted(local_cpus_of(local_rank))]
        k, n = peers.index(local_rank), len(peers)
        share = mine[k * len(mine) // n:(k + 1) * len(mine) // n]
        if share:
            return share, f"sysfs local_cpulist of the GPU's PCI device, share {k + 1}/{n} of its NUMA node"
    n = max(1, local_world)
    share = allowed[local_rank * len(allowed) // n:(local_rank + 1) * len(allowed) // n] or allowed
    return share, f"even split of the {len(allowed)} allowed CPUs by local rank (no NUMA information)"


def bind_rank_to_gpu_numa(local_rank, local_world, set_torch_threads=True, device_of=None):
    """One process per GPU: pin this process (and the threads it starts later: image decode, the CPU side of the launches) to the cores next to its GPU, so
    that 8 ranks do not migrate across sockets or pile onto the same cores.  No-op for a single local rank or where
    sched_setaffinity does not exist; RBA_NO_AFFINITY=1 switches it off.  device_of: local rank -> HIP device index (default: the rank itself; the share-device
    plumbing tests map every rank to device 0).  Returns a small record for the bench line."""
    rec = {"bound": False, "cpus": None, "source": None}
    if local_world <= 1 or os.environ.get("RBA_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return rec
    try:
        allowed = os.sched_getaffinity(0)
        dev_of = device_of or (lambda r: r)
        cache = {}

        def local_cpus(r):
            d = dev_of(r)
            if d not in cache:
                cache[d] = gpu_local_cpus(d) if torch.cuda.is_available() and d < torch.cuda.device_count() else None
            return cache[d]
        cpus, source = plan_rank_cpus(local_rank, local_world, local_cpus, allowed)
        os.sched_setaffinity(0, cpus)
        if set_torch_threads:
            torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))
        rec.update(bound=True, cpus=len(cpus), first_cpu=cpus[0], last_cpu=cpus[-1], source=source)
    except Exception as e:                                   # noqa: BLE001
        rec["source"] = f"not bound ({type(e).__name__}: {e})"
    return rec


def shard_indices(n_items: int, rank: int, world: int):
    """Image i -> rank i mod world (round-robin keeps shards balanced for any n)."""
    return list(range(rank, n_items, world))


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


_FORCE_COLLECTIVE = False


class force_collective:
    """with force_collective(): the exchanges below run their collectives even in a ONE-rank group (normally a 1-rank world returns its own tensor
    untouched) -- so that RCCL's all_gather / all_gather_into_tensor / all_reduce execute on a 1-GPU box (tests, `bench.py --rccl-one-rank`)."""

    def __enter__(self):
        global _FORCE_COLLECTIVE
        self.prev, _FORCE_COLLECTIVE = _FORCE_COLLECTIVE, True

    def __exit__(self, *exc):
        global _FORCE_COLLECTIVE
        _FORCE_COLLECTIVE = self.prev
        return False


def _skip_collective(world):
    return world == 1 and not (_FORCE_COLLECTIVE and dist.is_available() and dist.is_initialized())


@torch.no_grad()
def all_gather_variable(t: torch.Tensor) -> torch.Tensor:
    """Concatenate 1-d tensors of different lengths from all ranks (rank order): all_gather the sizes, pad to the
    max, one all_gather_into_tensor, trim."""
    world = _world()
    if _skip_collective(world):
        return t
    if t.is_cuda and dist.get_backend() == "gloo":      # gloo has no device all_gather: stage through the host (tests only)
        return all_gather_variable(t.cpu()).to(t.device)
    return _gather_padded(t, world)


def _gather_padded(t, world):
    """the collective body: sizes by all_gather, payload by ONE padded all_gather_into_tensor (device tensors on nccl = RCCL)"""
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if mx == 0:                                  # nobody has anything (every rank sees the same sizes, so every rank returns here)
        return t.new_empty(0)
    buf = t.new_zeros(mx)
    buf[: t.numel()] = t
    out = t.new_empty(world * mx)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)])


@torch.no_grad()
def pooled_ood_metrics(scores: torch.Tensor, labels: torch.Tensor) -> dict:
    """Exact metrics over the union of every rank's labelled pixels.  scores fp32 [n_r], labels bool/uint8 [n_r]
    (already restricted to labels in {0,1}).  Every rank returns the same dict."""
    from .metrics import ood_metrics

    s = all_gather_variable(scores.reshape(-1).float().contiguous())
    l = all_gather_variable(labels.reshape(-1).to(torch.uint8).contiguous())
    return ood_metrics(s, l)


@torch.no_grad()
def histogram_ood_metrics(scores: torch.Tensor, labels: torch.Tensor, bits: int = 16) -> dict:
    """Approximate pooled metrics from 2 x 2^bits int64 histograms over the order-preserving integer key of the fp32
    score (top `bits` bits): one 1 MB all_reduce instead of gathering every pixel.  Scores sharing a bin are treated
    as tied, so the result is the exact metric of the quantised scores (error <~ 1e-4 for RbA-range scores)."""
    from .metrics import ood_metrics

    s = scores.reshape(-1).float().contiguous()
    u = s.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    key = torch.where(u >= 0x80000000, 0xFFFFFFFF - u, u + 0x80000000) >> (32 - bits)     # monotone in s
    nb = 1 << bits
    pos = labels.reshape(-1).to(torch.bool)
    hist = torch.stack([torch.bincount(key[~pos], minlength=nb), torch.bincount(key[pos], minlength=nb)])
    if not _skip_collective(_world()):
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    nz = torch.nonzero(hist.sum(0)).reshape(-1)
    cnt = hist[:, nz]
    # expand to weighted curve: reuse ood_metrics by emitting each occupied bin once per class with multiplicity
    # through cumulative counts (descending key)
    neg, posc = cnt[0].flip(0), cnt[1].flip(0)
    tps, fps = torch.cumsum(posc, 0), torch.cumsum(neg, 0)
    return _metrics_from_curve(fps, tps)


def _metrics_from_curve(fps, tps):
    P, Nn = tps[-1].double(), fps[-1].double()
    tpd, fpd = tps.double(), fps.double()
    precision = tpd / (tpd + fpd)
    recall = tpd / P
    prev = torch.cat([recall.new_zeros(1), recall[:-1]])
    aupr = torch.sum((recall - prev) * precision)
    if fps.numel() > 2:
        d2f = fps[2:] - 2 * fps[1:-1] + fps[:-2]
        d2t = tps[2:] - 2 * tps[1:-1] + tps[:-2]
        one = torch.ones(1, dtype=torch.bool, device=fps.device)
        keep = torch.cat([one, (d2f != 0) | (d2t != 0), one])
        fps, tps = fps[keep], tps[keep]
    fpr = torch.cat([fps.new_zeros(1), fps]).double() / Nn
    tpr = torch.cat([tps.new_zeros(1), tps]).double() / P
    auroc = torch.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) * 0.5)
    above = torch.nonzero(tpr > 0.95).reshape(-1)
    fpr95 = fpr[above[0]] if above.numel() else fpr.new_zeros(())
    return {"auroc": float(auroc), "aupr": float(aupr), "fpr95": float(fpr95)}
