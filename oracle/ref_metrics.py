"""ORACLE -- test infrastructure, not product code.

OoD metrics exactly as the reference computes them (support.py:247-303): pixels with label 1
are OoD, label 0 inliers, everything else ignored; AuPRC = sklearn average_precision_score,
AUROC = sklearn roc_curve (default drop_intermediate=True) + auc, FPR95 = fpr at the first
retained ROC point with tpr > 0.95 (strict).  Pinned by tests/golden/g6_metrics.npz.
"""
import numpy as np


def evaluate_ood(anomaly_score: np.ndarray, ood_gts: np.ndarray) -> dict:
    from sklearn.metrics import roc_curve, auc, average_precision_score

    ood_gts = ood_gts.squeeze()
    anomaly_score = anomaly_score.squeeze()
    ood_out = anomaly_score[ood_gts == 1]
    ind_out = anomaly_score[ood_gts == 0]
    val_out = np.concatenate((ind_out, ood_out))
    val_label = np.concatenate((np.zeros(len(ind_out)), np.ones(len(ood_out))))
    aupr = average_precision_score(val_label, val_out)
    fpr, tpr, thr = roc_curve(val_label, val_out)
    roc_auc = auc(fpr, tpr)
    fpr_best = 0
    for i, j, k in zip(tpr, fpr, thr):
        if i > 0.95:
            fpr_best = j
            break
    return {"auroc": roc_auc, "aupr": aupr, "fpr95": fpr_best}
