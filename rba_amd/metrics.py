"""AUROC / AuPRC / FPR@95 over pooled pixels, computed with torch on whatever device the scores live on (the GPU
in the product) and matching what the reference gets from scikit-learn (support.py:247-268):

* AuPRC  = average_precision_score = sum_n (R_n - R_{n-1}) P_n over distinct thresholds, descending score;
* AUROC  = auc(roc_curve(...)) -- trapezoid over the ROC polyline;
* FPR95  = fpr of the first ROC point with tpr > 0.95 (strict) on the curve roc_curve returns with its default
  ``drop_intermediate=True`` (collinear interior points removed: keep i iff the second difference of fps or tps
  around i is non-zero, plus both end points, plus the prepended (0, 0)).

One global descending sort (rocPRIM radix sort on the GPU), integer cumsums, float64 ratios.
"""
import torch


def _sorted_curve(scores: torch.Tensor, labels: torch.Tensor):
    """-> (fps, tps, thresholds) at each distinct threshold, descending score order.  One key-value sort (the order inside a run of equal scores
    does not matter: the counts are read at the LAST index of every run), the cumulative positives in int32 while they fit -- at 288 images of
    1024 x 2048 this is 6 x 10^8 pairs, and an int64 gather + cumsum of them was a third of the metric's time (round 4)."""
    scores = scores.reshape(-1)
    labels = labels.reshape(-1)
    s, order = torch.sort(scores, descending=True)
    n = s.numel()
    small = n < 2 ** 31 - 1
    y = labels[order].to(torch.int32 if small else torch.int64)
    del order
    distinct = torch.nonzero(s[1:] != s[:-1]).reshape(-1)
    idx = torch.cat([distinct, torch.tensor([n - 1], device=s.device, dtype=distinct.dtype)])
    tps = torch.cumsum(y, 0, dtype=y.dtype)[idx].to(torch.int64)
    fps = 1 + idx - tps
    return fps, tps, s[idx]


@torch.no_grad()
def binary_clf_curve(scores: torch.Tensor, labels: torch.Tensor):
    """fps, tps (int64) at each distinct threshold, descending score order."""
    fps, tps, _ = _sorted_curve(scores, labels)
    return fps, tps


@torch.no_grad()
def ood_metrics(scores: torch.Tensor, labels: torch.Tensor) -> dict:
    """scores: float tensor, labels: {0,1} tensor of the same number of elements (1 = OoD = positive)."""
    if scores.numel() != labels.numel():
        raise ValueError("scores and labels differ in size")
    if scores.numel() == 0:
        raise ValueError("no labelled pixels")
    fps, tps = binary_clf_curve(scores, labels)
    P, Nn = tps[-1].double(), fps[-1].double()
    # ---- average precision
    tpd, fpd = tps.double(), fps.double()
    precision = tpd / (tpd + fpd)
    recall = tpd / P
    prev = torch.cat([recall.new_zeros(1), recall[:-1]])
    aupr = torch.sum((recall - prev) * precision)
    # ---- ROC with drop_intermediate
    if fps.numel() > 2:
        d2f = fps[2:] - 2 * fps[1:-1] + fps[:-2]
        d2t = tps[2:] - 2 * tps[1:-1] + tps[:-2]
        keep = torch.cat([torch.ones(1, dtype=torch.bool, device=fps.device), (d2f != 0) | (d2t != 0),
                          torch.ones(1, dtype=torch.bool, device=fps.device)])
        fps, tps = fps[keep], tps[keep]
    fpr = torch.cat([fps.new_zeros(1), fps]).double() / Nn
    tpr = torch.cat([tps.new_zeros(1), tps]).double() / P
    auroc = torch.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) * 0.5)
    above = torch.nonzero(tpr > 0.95).reshape(-1)
    fpr95 = fpr[above[0]] if above.numel() else fpr.new_zeros(())
    return {"auroc": float(auroc), "aupr": float(aupr), "fpr95": float(fpr95)}


@torch.no_grad()
def roc_at_tpr95(scores: torch.Tensor, labels: torch.Tensor):
    """OODEvaluator.calculate_auroc (support.py:247-257): (auc of roc_curve, fpr and threshold of the first ROC point with
    tpr > 0.95; if there is none: fpr 0 and the last threshold).  roc_curve's first threshold is +inf (scikit-learn >= 1.3)."""
    fps, tps, thr = _sorted_curve(scores, labels)
    P, Nn = tps[-1].double(), fps[-1].double()
    if fps.numel() > 2:
        d2f = fps[2:] - 2 * fps[1:-1] + fps[:-2]
        d2t = tps[2:] - 2 * tps[1:-1] + tps[:-2]
        one = torch.ones(1, dtype=torch.bool, device=fps.device)
        keep = torch.cat([one, (d2f != 0) | (d2t != 0), one])
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    fpr = torch.cat([fps.new_zeros(1), fps]).double() / Nn
    tpr = torch.cat([tps.new_zeros(1), tps]).double() / P
    thr = torch.cat([thr.new_full((1,), float("inf")), thr])
    auroc = torch.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) * 0.5)
    above = torch.nonzero(tpr > 0.95).reshape(-1)
    if above.numel():
        return float(auroc), float(fpr[above[0]]), float(thr[above[0]])
    return float(auroc), 0, float(thr[-1])


@torch.no_grad()
def select_labelled(anomaly_score: torch.Tensor, ood_gts: torch.Tensor):
    """Keep pixels labelled 0 (inlier) or 1 (OoD); everything else (255) is ignored (support.py:275-285)."""
    s = anomaly_score.reshape(-1)
    g = ood_gts.reshape(-1)
    idx = torch.nonzero((g == 0) | (g == 1)).reshape(-1)          # one compaction, two gathers
    return s[idx], (g[idx] == 1)
