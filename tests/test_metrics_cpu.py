"""maest_amd.metrics against scikit-learn (the reference's own metric calls,
models/module.py:181-182), including tied scores."""
import numpy as np
import torch
from sklearn import metrics as skm

from maest_amd import metrics as M


def _data(n, c, seed, ties=False):
    rng = np.random.Generator(np.random.PCG64(seed))
    y = (rng.random((n, c)) < 0.2).astype(np.float32)
    y[0], y[1] = 1.0, 0.0                                     # every class has both labels
    s = rng.random((n, c)).astype(np.float32)
    if ties:
        s = np.round(s * 8) / 8                               # many equal scores
    return y, s


def test_macro_ap_and_roc_match_sklearn():
    for ties in (False, True):
        y, s = _data(300, 17, 5, ties)
        ap = M.macro_average_precision(torch.from_numpy(y), torch.from_numpy(s))
        roc = M.macro_roc_auc(torch.from_numpy(y), torch.from_numpy(s))
        assert abs(ap - skm.average_precision_score(y, s, average="macro")) < 1e-9, ties
        assert abs(roc - skm.roc_auc_score(y, s, average="macro")) < 1e-9, ties
    y, s = _data(64, 400, 6)
    per = M.average_precision(torch.from_numpy(y), torch.from_numpy(s)).numpy()
    want = skm.average_precision_score(y, s, average=None)
    assert np.allclose(per, want, atol=1e-9)
