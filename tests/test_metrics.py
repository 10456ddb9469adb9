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
