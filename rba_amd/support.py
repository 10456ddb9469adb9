"""``OODEvaluator`` with the reference's constructor and method signatures (support.py:228-399): per-image scoring
loop, then AUROC / AuPRC / FPR95 over all labelled pixels.  The metrics are computed by ``rba_amd.metrics`` (GPU
sort) instead of scikit-learn; results match to float64 round-off (tests/test_metrics.py)."""
from typing import Callable

import numpy as np
import torch

from . import ops
from .h2d import to_device
from .metrics import ood_metrics, select_labelled


class _HostRing:
    """Device -> host copies through `depth` reusable pinned staging buffers: push() enqueues an asynchronous copy and returns; the
    slot's previous content is delivered (copied into a pageable numpy array, handed to its sink) when the slot comes round again or
    at drain().  Page-locked memory: `depth` maps, whatever the number of images."""

    def __init__(self, depth=4):
        self.depth = depth
        self.slots = [None] * depth             # [pinned buffer, event, sink | None]
        self.n = 0

    def _deliver(self, slot):
        buf, ev, sink = slot
        if sink is not None:
            ev.synchronize()
            sink(np.array(buf.numpy(), copy=True))
            slot[2] = None

    def push(self, t, sink):
        i = self.n % self.depth
        self.n += 1
        slot = self.slots[i]
        if slot is not None:
            self._deliver(slot)
        if slot is None or slot[0].shape != t.shape or slot[0].dtype != t.dtype:
            slot = self.slots[i] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True), torch.cuda.Event(), None]
        slot[0].copy_(t, non_blocking=True)
        slot[1].record(torch.cuda.current_stream(t.device))
        slot[2] = sink

    def drain(self):
        for k in range(self.depth):             # oldest first
            slot = self.slots[(self.n + k) % self.depth]
            if slot is not None:
                self._deliver(slot)


class OODEvaluator:
    def __init__(self, model, inference_func: Callable, anomaly_score_func: Callable):
        self.model = model
        self.inference_func = inference_func
        self.anomaly_score_func = anomaly_score_func

    def get_logits(self, x, **kwargs):
        return self.inference_func(self.model, x, **kwargs)

    def get_anomaly_score(self, x, **kwargs):
        return self.anomaly_score_func(self.model, x, **kwargs)

    def calculate_auroc(self, conf, gt):
        """-> (auroc, fpr at the first tpr > 0.95, the threshold there) (support.py:247-257)."""
        from .metrics import roc_at_tpr95
        dev = getattr(self.model, "device", torch.device("cpu")) if self.model is not None else torch.device("cpu")
        return roc_at_tpr95(torch.as_tensor(conf, device=dev), torch.as_tensor(gt, device=dev))

    def calculate_ood_metrics(self, out, label):
        """-> (auroc, aupr, fpr95) for flat score / {0,1} label arrays (support.py:259-268)."""
        dev = getattr(self.model, "device", torch.device("cpu")) if self.model is not None else torch.device("cpu")
        r = ood_metrics(torch.as_tensor(out, device=dev), torch.as_tensor(label, device=dev))
        return r["auroc"], r["aupr"], r["fpr95"]

    def evaluate_ood(self, anomaly_score, ood_gts, verbose=True):
        """anomaly_score [N,H,W], ood_gts [N,1,H,W] (numpy or torch); labels 1 = OoD, 0 = inlier, else ignored."""
        dev = getattr(self.model, "device", torch.device("cpu")) if self.model is not None else torch.device("cpu")
        s = torch.as_tensor(np.asarray(anomaly_score) if not torch.is_tensor(anomaly_score) else anomaly_score).to(dev)
        g = torch.as_tensor(np.asarray(ood_gts) if not torch.is_tensor(ood_gts) else ood_gts).to(dev)
        s, y = select_labelled(s.squeeze(), g.squeeze())
        if verbose:
            print(f"Calculating Metrics for {s.numel()} Points ...")
        result = ood_metrics(s, y)
        if verbose:
            print(f"Max Logits: AUROC score: {result['auroc']}")
            print(f"Max Logits: AUPRC score: {result['aupr']}")
            print(f"Max Logits: FPR@TPR95: {result['fpr95']}")
        return result

    def compute_anomaly_scores(self, loader, device=torch.device("cpu"), return_preds=False,
                               use_gaussian_smoothing=False, upper_limit=450):
        """Batch-1 scoring loop (support.py:353-399).  Returns numpy arrays like the reference.

        Device -> host: the reference's `.cpu()` per image (support.py:375, 390) stalls the stream once per image; here every map goes
        through a small ring of reusable pinned staging buffers (`_HostRing`: 4 slots, events) into its final pageable numpy array, so
        the launch thread never waits for the image it has just issued and page-locked memory stays at a few maps.
        NaN scores: the f16x3 token Linears answer |value| >= 65504 with NaN (never a wrong number, never +-inf); once per CHUNK of images a
        fused `isnan` flag per image is read back, and an image whose score holds a NaN is scored again on the full-range bf16x6
        kernels before anything reaches the rank statistics (FloatingPointError if that one holds a NaN too).  An INFINITE score of a custom
        score function is passed on, as the reference passes it on.  The re-score runs inside `ops.split_mode("bf16x6")`: the
        mode is per THREAD (round 5), other threads of the process keep scoring on their own mode."""
        anomaly_score, ood_gts, predictions = [], [], []
        on_gpu = torch.device(device).type == "cuda"
        if on_gpu:
            from .datasets import threaded
            loader = threaded(loader)                                           # worker processes -> decode threads, same batches (datasets.threaded)
        ring = _HostRing(4) if on_gpu else None
        mode = getattr(self.anomaly_score_func, "rba_score_mode", None)         # set on rba_amd.evaluate_ood's score functions
        CHUNK = 16
        chunk = []                                                              # (index, x, finite flag) of scores not yet checked
        self.bf16x6_rescored_images = []

        def score_one(x):
            preds = None
            if return_preds and mode is not None and hasattr(self.model, "rba_scores"):
                # one forward instead of the reference's two (support.py:380,386)
                score, preds = self.model.rba_scores([{"image": x[0]}], return_argmax=True, score=mode)[0]
                preds = preds.to(torch.int64).unsqueeze(0)
            else:
                score = self.get_anomaly_score(x)
                if return_preds:
                    preds = self.get_logits(x)[:, :19].max(dim=1)[1]
            if use_gaussian_smoothing:                                          # transforms.GaussianBlur(7, sigma=1), support.py:366-383
                score = ops.gaussian_blur(score.contiguous(), 7, 1.0)
            return score, preds

        def store(lst, idx, t):
            while len(lst) <= idx:
                lst.append(None)
            if ring is not None and t.is_cuda:
                ring.push(t, lambda a, lst=lst, idx=idx: lst.__setitem__(idx, a))
            else:
                lst[idx] = t.cpu().numpy()

        def check_chunk():
            if not chunk:
                return
            has_nan = torch.stack([c[2] for c in chunk]).cpu()                  # one read-back per chunk
            for (idx, x, _), bad in zip(chunk, has_nan.tolist()):
                if not bad:
                    continue
                prev = getattr(self.model, "graph_replay", None)
                with ops.split_mode("bf16x6"):
                    if prev is not None:
                        self.model.graph_replay = False
                    try:
                        score, preds = score_one(x)
                    finally:
                        if prev is not None:
                            self.model.graph_replay = prev
                if bool(torch.isnan(score).any()):
                    raise FloatingPointError(f"image {idx}: NaN anomaly score in the f16x3 AND the bf16x6 arithmetic")
                self.bf16x6_rescored_images.append(idx)
                store(anomaly_score, idx, score)
                if return_preds:
                    store(predictions, idx, preds)
            chunk.clear()

        for jj, (x, y) in enumerate(loader):
            if jj >= upper_limit:
                break
            x = to_device(x, device)                                             # never a DMA from a DataLoader worker's shared-memory pages (h2d.py)
            ood_gts.append(np.asarray(y.cpu()))
            score, preds = score_one(x)
            store(anomaly_score, jj, score)
            if return_preds:
                store(predictions, jj, preds)
            if score.is_cuda:
                chunk.append((jj, x, torch.isnan(score).any()))
                if len(chunk) >= CHUNK:
                    check_chunk()
            elif bool(torch.isnan(score).any()):
                raise FloatingPointError(f"image {jj}: NaN anomaly score")
        check_chunk()
        if ring is not None:
            ring.drain()
        ood_gts = np.array(ood_gts)
        anomaly_score = np.array(anomaly_score)
        if return_preds:
            return anomaly_score, ood_gts, np.array(predictions)
        return anomaly_score, ood_gts

    def evaluate_ood_bootstrapped(self, dataset, ratio, trials, device=torch.device("cpu"), batch_size=1, num_workers=10):
        """Metrics over `trials` random subsets of ratio * len(dataset) images -> (means, stds) in percent
        (support.py:305-351; np.random.choice without replacement, as there)."""
        from torch.utils.data import DataLoader, Subset
        results = {}
        n = len(dataset)
        for _ in range(trials):
            idx = np.random.choice(np.arange(n), int(n * ratio), replace=False)
            loader = DataLoader(Subset(dataset, idx.tolist()), batch_size=batch_size, num_workers=num_workers)
            score, gts = self.compute_anomaly_scores(loader=loader, device=device, return_preds=False)
            for k, v in self.evaluate_ood(score, gts, verbose=False).items():
                results.setdefault(k, []).append(v)
        means = {k: float(np.mean(v)) * 100.0 for k, v in results.items()}
        stds = {k: float(np.std(v)) * 100.0 for k, v in results.items()}
        return means, stds
