"""Evaluation metrics of the reference's validation loop on the device.

Reference: ``Module.on_test_validation_epoch_end`` models/module.py:156-202 gathers ``y`` and ``sigmoid(logits)``
over the epoch (and over ranks), moves them to the host and calls scikit-learn's
``average_precision_score(y, y_hat, average="macro")`` and ``roc_auc_score(y, y_hat, average="macro")``.
Here both are computed where the predictions already are, per class in parallel, with sort / cumsum (eval-side
glue, not the hot path); ties in the scores are handled as scikit-learn does (one threshold per distinct score;
average ranks).  Pinned against scikit-learn in tests/test_metrics_cpu.py.
"""
from __future__ import annotations

import torch


def _sorted_by_score(y: torch.Tensor, s: torch.Tensor):
    order = torch.argsort(s, dim=0, descending=True, stable=True)
    return torch.gather(y, 0, order), torch.gather(s, 0, order)


def average_precision(y: torch.Tensor, y_hat: torch.Tensor) -> torch.Tensor:
    """Per-class AP, [C].  y: {0,1} [N, C]; y_hat: scores [N, C]."""
    y = y.to(torch.float64)
    ys, ss = _sorted_by_score(y, y_hat.to(torch.float64))
    tp = torch.cumsum(ys, 0)
    k = torch.arange(1, y.shape[0] + 1, device=y.device, dtype=torch.float64).unsqueeze(1)
    # a threshold sits at the LAST element of every run of equal scores
    last = torch.ones_like(ss, dtype=torch.bool)
    last[:-1] = ss[:-1] != ss[1:]
    prec = tp / k
    npos = tp[-1].clamp_min(1e-300)
    tp_at = torch.where(last, tp, torch.zeros_like(tp))
    # recall increments between consecutive thresholds: tp at this threshold minus tp at the previous one
    prev = torch.cummax(torch.where(last, tp, torch.full_like(tp, -1.0)), 0).values
    prev = torch.cat([torch.zeros_like(prev[:1]), prev[:-1]], 0).clamp_min(0.0)
    dr = torch.where(last, (tp_at - prev) / npos, torch.zeros_like(tp))
    return (dr * prec).sum(0)


def roc_auc(y: torch.Tensor, y_hat: torch.Tensor) -> torch.Tensor:
    """Per-class ROC-AUC, [C], by the rank statistic with average ranks for ties."""
    y = y.to(torch.float64)
    s = y_hat.to(torch.float64)
    N = y.shape[0]
    order = torch.argsort(s, dim=0, stable=True)
    ss = torch.gather(s, 0, order)
    ranks_sorted = torch.arange(1, N + 1, device=y.device, dtype=torch.float64).unsqueeze(1).expand_as(ss).clone()
    # average the ranks inside every run of equal scores: run id -> (first, last) position
    new_run = torch.ones_like(ss, dtype=torch.bool)
    new_run[1:] = ss[1:] != ss[:-1]
    run_id = torch.cumsum(new_run.to(torch.int64), 0) - 1
    pos = torch.arange(N, device=y.device, dtype=torch.float64).unsqueeze(1).expand_as(ss)
    first = torch.zeros_like(ss).scatter_reduce(0, run_id, pos, "amin", include_self=False)
    lastp = torch.zeros_like(ss).scatter_reduce(0, run_id, pos, "amax", include_self=False)
    avg_rank = (torch.gather(first, 0, run_id) + torch.gather(lastp, 0, run_id)) / 2.0 + 1.0
    ranks = torch.empty_like(avg_rank).scatter_(0, order, avg_rank)
    npos = y.sum(0)
    nneg = N - npos
    u = (ranks * y).sum(0) - npos * (npos + 1.0) / 2.0
    return u / (npos * nneg)


def macro_average_precision(y, y_hat) -> float:
    return float(average_precision(y, y_hat).mean())


def macro_roc_auc(y, y_hat) -> float:
    return float(roc_auc(y, y_hat).mean())
