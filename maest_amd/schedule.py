"""Learning-rate schedules of the training loop (host logic; reference ``helpers/ramp.py`` as used by
``Module.get_scheduler_lambda`` / ``get_lr_scheduler``, models/module.py:213-235).  Each factory returns
``epoch -> factor`` for ``torch.optim.lr_scheduler.LambdaLR``; values are pinned to the reference by
``tests/golden/g10_lr_schedule.npz`` (oracle/gen_golden_schedule.py)."""
from __future__ import annotations

import numpy as np


def exp_warmup_linear_down(warmup, rampdown_length, start_rampdown, last_value):
    """exp(-5 (1 - e / warmup)^2) for the first ``warmup`` epochs (the epoch clipped to >= 0.5), times a hold at 1 until
    ``start_rampdown`` followed by a straight line down to ``last_value`` over ``rampdown_length`` epochs."""
    def factor(epoch):
        up = 1.0
        if epoch < warmup:
            rest = 1.0 - np.clip(epoch, 0.5, warmup) / warmup
            up = float(np.exp(-5.0 * rest * rest))
        since = epoch - start_rampdown
        if since <= 0:
            down = 1.0
        elif since < rampdown_length:
            down = last_value + (1.0 - last_value) * (rampdown_length - since) / rampdown_length
        else:
            down = last_value
        return up * down
    return factor


def cosine_cycle(cycle_len=20, ramp_down_start=100, last_lr_value=0.01):
    """Cosine cycles of ``cycle_len`` epochs between 1 and ``last_lr_value`` (starting half a cycle in, i.e. at the
    bottom for even lengths), flat at ``last_lr_value`` once the epoch passes ``ramp_down_start`` rounded up to the end
    of its cycle."""
    stop = cycle_len + (ramp_down_start - 1) // cycle_len * cycle_len

    def factor(epoch):
        if epoch > stop:
            return last_lr_value
        phase = (epoch + cycle_len // 2.0) / (1.0 * cycle_len)
        return float(last_lr_value + (1.0 - last_lr_value) * 0.5 * (np.cos(2.0 * np.pi * phase) + 1))
    return factor


def scheduler_lambda(schedule_mode="exp_lin", warm_up_len=5, ramp_down_start=50, ramp_down_len=50, last_lr_value=0.01):
    """``Module.get_scheduler_lambda`` (models/module.py:213-226) with the reference's defaults (:31-41)."""
    if schedule_mode == "exp_lin":
        return exp_warmup_linear_down(warm_up_len, ramp_down_len, ramp_down_start, last_lr_value)
    if schedule_mode == "cos_cyc":
        return cosine_cycle(warm_up_len, ramp_down_start, last_lr_value)
    raise RuntimeError(f"schedule_mode={schedule_mode} Unknown for a lambda funtion.")
