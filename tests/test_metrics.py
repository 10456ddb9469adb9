"""CPU: product metrics (rba_amd.metrics, torch) against the sklearn-based oracle and the golden values."""
import numpy as np
import pytest
import torch

from oracle import ref_metrics
from rba_amd import metrics
from rba_amd.support import OODEvaluator


@pytest.mark.parametrize("case", "abcd")
def test_matches_golden(golden, case):
    g = golden("g6_metrics")
    r = OODEvaluator(None, None, None).evaluate_ood(g[case + "_score"], g[case + "_gt"], verbose=False)
    np.testing.assert_allclose([r["auroc"], r["aupr"], r["fpr95"]], g[case + "_metrics"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("seed", range(6))
def test_matches_sklearn_random(seed):
    rng = np.random.RandomState(seed)
    n = int(rng.randint(50, 5000))
    gt = rng.choice([0, 1, 255], size=n, p=[0.8, 0.1, 0.1])
    gt[:2] = [0, 1]
    score = (rng.randn(n) + (gt == 1) * rng.rand() * 2).astype(np.float32)
    if seed % 2:
        score = np.round(score, 1)      # ties
    want = ref_metrics.evaluate_ood(score, gt)
    s, y = metrics.select_labelled(torch.from_numpy(score), torch.from_numpy(gt))
    got = metrics.ood_metrics(s, y)
    for k in want:
        assert abs(got[k] - want[k]) < 1e-12, (k, got[k], want[k])


def test_fpr95_zero_when_first_point_passes():
    # every positive above every negative: first retained point already has tpr = 1 > 0.95 at fpr 0
    s = torch.tensor([3.0, 2.5, 0.1, 0.2, 0.3])
    y = torch.tensor([1, 1, 0, 0, 0])
    r = metrics.ood_metrics(s, y)
    assert r == {"auroc": 1.0, "aupr": 1.0, "fpr95": 0.0}


@pytest.mark.parametrize("seed", range(4))
def test_calculate_auroc_matches_sklearn(seed):
    """OODEvaluator.calculate_auroc (support.py:247-257): (auc, fpr, threshold) at the first ROC point with tpr > 0.95."""
    from sklearn.metrics import auc, roc_curve
    rng = np.random.RandomState(100 + seed)
    n = 3000
    gt = rng.choice([0, 1], size=n, p=[0.9, 0.1])
    conf = (rng.randn(n) + 1.5 * gt).astype(np.float32)
    if seed % 2:
        conf = np.round(conf, 1)
    fpr, tpr, thr = roc_curve(gt, conf)
    want = None
    for i, j, k in zip(tpr, fpr, thr):
        if i > 0.95:
            want = (auc(fpr, tpr), j, k)
            break
    got = OODEvaluator(None, None, None).calculate_auroc(conf, gt)
    assert abs(got[0] - want[0]) < 1e-12 and abs(got[1] - want[1]) < 1e-12 and got[2] == pytest.approx(float(want[2]))


def test_evaluate_ood_bootstrapped_shapes():
    """means / stds in percent over random image subsets (support.py:305-351), with a stand-in score function on CPU"""
    rng = np.random.RandomState(0)
    data = []
    for _ in range(8):
        gt = torch.from_numpy(rng.choice([0, 1, 255], size=(12, 16), p=[0.8, 0.1, 0.1]))
        data.append((torch.from_numpy(rng.randn(3, 12, 16).astype(np.float32)) + (gt == 1) * 2.0, gt))
    ev = OODEvaluator(None, None, lambda model, x: x[0].mean(0))
    np.random.seed(0)
    means, stds = ev.evaluate_ood_bootstrapped(data, ratio=0.5, trials=3, num_workers=0)
    assert set(means) == {"auroc", "aupr", "fpr95"} == set(stds)
    assert 50.0 < means["auroc"] <= 100.0 and all(v >= 0 for v in stds.values())
