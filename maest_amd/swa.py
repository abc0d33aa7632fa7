"""Stochastic weight averaging of the MAEST weights on the device.

Reference: ``StochasticWeightAveragingAndCopy`` helpers/swa_callback.py:9-44 = Lightning's
``StochasticWeightAveraging`` (running mean ``avg += (w - avg) / (n + 1)`` once per epoch from ``swa_epoch_start``)
whose averaged model is copied into ``pl_module.net_swa`` at every epoch end; checkpoints then carry ``net_swa.*``
keys, which ``get_maest(checkpoint=..., checkpoint_swa_weigts=True)`` selects (models/maest.py:1554-1567).
Here the running mean of all 85.9 M parameters is ONE kernel launch (csrc/misc.hip: swa_update_kernel).
"""
from __future__ import annotations

import torch

from . import ops


class WeightAverager:
    def __init__(self, net: torch.nn.Module):
        self.net = net
        # what the reference stores as pl_module.net_swa.  Built from a fresh constructor + the current weights, never
        # by copying the live object: SWA starts MID-training (swa_epoch_start), when `net` carries engine state
        # (operand-copy caches, streams, captured graphs, the GradReducer's flat buffer and process group)
        self.net_swa = net.clone_weights() if hasattr(net, "clone_weights") else __import__("copy").deepcopy(net)
        for p in self.net_swa.parameters():
            p.requires_grad_(False)
        self.n_averaged = 0

    @torch.no_grad()
    def update(self):
        """Call once per epoch (or per `swa_freq` steps): fold the current weights into the running mean."""
        cur = [p.detach() for p in self.net.parameters()]
        avg = [p.detach() for p in self.net_swa.parameters()]
        if self.n_averaged == 0:                        # first call: avg <- w  (inv_count = 1)
            ops.swa_update_multi(avg, cur, 1.0)
        else:
            ops.swa_update_multi(avg, cur, 1.0 / (self.n_averaged + 1))
        self.n_averaged += 1
        if hasattr(self.net_swa, "_engine"):            # the kernel writes in place behind torch's version counters:
            self.net_swa._engine.w.clear()              # drop the averaged model's cached operand copies
        return self.net_swa

    def state_dict(self, prefix_net="net.", prefix_swa="net_swa."):
        """The Lightning-checkpoint layout the reference's get_maest(checkpoint=...) reads."""
        sd = {prefix_net + k: v for k, v in self.net.state_dict().items()}
        sd.update({prefix_swa + k: v for k, v in self.net_swa.state_dict().items()})
        return {"state_dict": sd}
