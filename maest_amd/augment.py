"""Host-side random draws of the training-time augmentations (the arithmetic itself is fused into the HIP kernels:
mixup into csrc/embed.hip:patch_im2col_kernel and csrc/misc.hip:bce_logits_kernel)."""
import numpy as np
import torch


def mixup_draw(batch_size: int, alpha: float):
    """Partner permutation and mixing weights of one mixup batch.

    Consumes the RNG streams exactly like the reference's ``my_mixup`` (helpers/mixup.py:5-12): first one
    ``torch.randperm(batch_size)`` from torch's global generator, then ``batch_size`` Beta(alpha, alpha) variates
    from numpy's global generator; the weight of a clip is the larger of (beta, 1 - beta), evaluated in float32.
    Returns (int64 permutation [B], float32 weights [B])."""
    partner = torch.randperm(batch_size)
    beta = np.random.beta(alpha, alpha, batch_size).astype(np.float32)
    weight = np.maximum(beta, np.float32(1.0) - beta)
    return partner, torch.from_numpy(weight)


my_mixup = mixup_draw   # the reference's name for it
