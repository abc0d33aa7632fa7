"""SpecAugment-style masking (reference: helpers/spec_masking.py:4-33, built on torchaudio's
TimeMasking(time_mask_param=8, iid_masks=True, p=0.2) x20 and FrequencyMasking(freq_mask_param=5) x8).

The stripes are SAMPLED on the host with torchaudio's documented rule (``mask_along_axis_iid``:
width = U[0,1) * param', start = U[0,1) * (size - width), param' = min(param, floor(size * p)) when
p < 1; mask where floor(start) <= idx < floor(start) + floor(width)... see SURVEY 8c -- parity with a
particular torchaudio version is unpinned) and APPLIED on the device by csrc/embed.hip:spec_mask_kernel.
"""
import torch

from . import ops


class SpecMasking:
    def __init__(self, time_mask_param=8, freq_mask_param=5, p=0.2, iid_masks=True, time_masks=20, freq_masks=8):
        self.time_mask_param = time_mask_param
        self.freq_mask_param = freq_mask_param
        self.p = p
        self.iid_masks = iid_masks
        self.time_masks = time_masks
        self.freq_masks = freq_masks

    @staticmethod
    def _draw(n_masks, batch, size, param, p):
        if p < 1.0:
            param = min(param, int(size * p))
        if param < 1 or n_masks == 0:
            return torch.zeros((batch, max(n_masks, 0), 2), dtype=torch.int32)
        width = torch.rand(batch, n_masks) * param
        start = torch.rand(batch, n_masks) * (size - width)
        lo = start.long()
        hi = lo + width.long()
        return torch.stack([lo, hi - lo], dim=-1).to(torch.int32)

    def draw(self, batch, n_freq, n_time):
        """-> (t_stripes [B, time_masks, 2], f_stripes [B, freq_masks, 2]) int32 (start, width)."""
        t = self._draw(self.time_masks, batch, n_time, self.time_mask_param, self.p)
        f = self._draw(self.freq_masks, batch, n_freq, self.freq_mask_param, 1.0)
        return t, f

    def compute(self, batch: torch.Tensor) -> torch.Tensor:
        """batch: [..., F, T] fp32 on the device; masked in place and returned."""
        shape = batch.shape
        x = batch.reshape(-1, shape[-2], shape[-1])
        t, f = self.draw(x.shape[0], shape[-2], shape[-1])
        ops.spec_mask_(x, t.to(x.device).contiguous(), f.to(x.device).contiguous())
        return batch
