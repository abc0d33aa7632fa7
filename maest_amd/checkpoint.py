"""Checkpoint interop for the MAEST weights: host-side plumbing at LOAD time, plain torch.

Reference being mirrored (paths relative to palonso/MAEST):
  * ``get_maest(checkpoint=...)`` models/maest.py:1554-1567 -- Lightning ``.ckpt`` -> ``state_dict``, ``net_swa.`` /
    ``net.`` prefixes, optional head discard, ``strict=False``                          -> load_lightning_checkpoint
  * ``checkpoint_filter_fn`` models/maest.py:1051-1118 -- DeiT/ImageNet ``pos_embed`` -> (cls/dist, freq, time)
    embeddings (``adapt_image_pos_embed_to_passt`` :1010-1034), MAEST -> MAEST with another input size
    (``adapt_passt_timefreq_embed`` :1037-1048, bicubic), old linear patchify weights   -> adapt_state_dict
  * packaging/push_to_hub.py:30-100 -- MAEST -> HF AudioSpectrogramTransformer key names, fused QKV split,
    freq + time position tables recombined into one                                      -> to_hf_ast_state_dict
    (and the inverse, from_hf_ast_state_dict, so that weights published in that layout run on this engine).
The tensors produced here are ordinary fp32 parameters; the hot path never sees any of this code.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


def adapt_image_pos_embed(posemb: torch.Tensor, num_tokens: int, gs_new: Tuple[int, int], mode: str = "bicubic"):
    """ViT ``pos_embed`` [1, num_tokens + g*g, D] -> (token part, freq [1,D,F,1], time [1,D,1,T])."""
    tok = posemb[:, :num_tokens] if num_tokens else posemb[:, :0]
    grid = posemb[0, num_tokens:] if num_tokens else posemb[0]
    g = int(math.sqrt(len(grid)))
    grid = grid.reshape(1, g, g, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=list(gs_new), mode=mode, align_corners=False)
    return tok, grid.mean(dim=3, keepdim=True), grid.mean(dim=2, keepdim=True)


def adapt_timefreq_embed(freq: torch.Tensor, time: torch.Tensor, n_freq: int, n_time: int, mode: str = "bicubic"):
    """MAEST freq/time tables -> another patch grid (input size change, e.g. 10 s weights in a 30 s model)."""
    return (F.interpolate(freq, size=[n_freq, 1], mode=mode, align_corners=False),
            F.interpolate(time, size=[1, n_time], mode=mode, align_corners=False))


def adapt_state_dict(state_dict: Dict[str, torch.Tensor], model) -> Dict[str, torch.Tensor]:
    """``checkpoint_filter_fn``: make a pretrained state dict fit `model` (position tables, patchify weights)."""
    if "model" in state_dict:                       # DeiT release files
        state_dict = state_dict["model"]
    sd = dict(state_dict)
    n_freq, n_time = model.patch_embed.grid_size
    if "time_new_pos_embed" not in sd:              # ImageNet / DeiT checkpoint
        tok, freq, time = adapt_image_pos_embed(sd.pop("pos_embed"), getattr(model, "num_tokens", 1), (n_freq, n_time))
        sd["new_pos_embed"], sd["freq_new_pos_embed"], sd["time_new_pos_embed"] = tok, freq, time
    elif sd["freq_new_pos_embed"].shape[2] != n_freq or sd["time_new_pos_embed"].shape[3] != n_time:
        sd["freq_new_pos_embed"], sd["time_new_pos_embed"] = adapt_timefreq_embed(
            sd["freq_new_pos_embed"], sd["time_new_pos_embed"], n_freq, n_time)
    out = {}
    for k, v in sd.items():
        if "patch_embed.proj.weight" in k and v.dim() < 4:
            o, _, h, w = model.patch_embed.proj.weight.shape
            v = v.reshape(o, -1, h, w)
        out[k] = v
    return out


def load_lightning_checkpoint(model, path, swa_weights: bool = True, discard_head: bool = False, adapt: bool = False):
    """``get_maest(checkpoint=path)``; with ``adapt=True`` the position tables are re-interpolated first."""
    state_dict = torch.load(path, map_location="cpu")["state_dict"]
    prefix = "net_swa." if swa_weights else ""
    state_dict = {k.replace(prefix, ""): v for k, v in state_dict.items()}
    if discard_head:
        state_dict = {k: v for k, v in state_dict.items() if "head" not in k}
    if adapt:
        state_dict = adapt_state_dict(state_dict, model)
    return model.load_state_dict(state_dict, strict=False)


# ---- HF AudioSpectrogramTransformer layout (packaging/push_to_hub.py:30-100) -----------------------------
_AST = "audio_spectrogram_transformer."
_RENAMES = [("blocks.", _AST + "encoder.layer."), ("cls_token", _AST + "embeddings.cls_token"),
            ("dist_token", _AST + "embeddings.distillation_token"),
            ("patch_embed.proj.", _AST + "embeddings.patch_embeddings.projection."), ("norm.", _AST + "layernorm."),
            ("norm1.", "layernorm_before."), ("norm2.", "layernorm_after."), ("mlp.fc1.", "intermediate.dense."),
            ("mlp.fc2.", "output.dense."), ("attn.proj.", "attention.output.dense."),
            ("head.0.", "classifier.layernorm."), ("head.1.", "classifier.dense.")]


def to_hf_ast_state_dict(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    # head_dist is dropped like the reference does (push_to_hub.py:111-115); the mel front end's constant tables
    # (torchaudio buffers) are no AST weights either
    sd = {k: v for k, v in state_dict.items() if not k.startswith(("head_dist.", "melspectrogram."))}
    pos = (sd.pop("freq_new_pos_embed") + sd.pop("time_new_pos_embed")).flatten(2, 3).transpose(1, 2)
    sd[_AST + "embeddings.position_embeddings"] = torch.cat((sd.pop("new_pos_embed"), pos), dim=1)
    for a, b in _RENAMES:                            # same order as the reference: "norm." before "norm1."
        sd = {k.replace(a, b): v for k, v in sd.items()}
    out = {}
    for k, v in sd.items():
        if "qkv" in k:
            layer, kind = k.split(".")[3], k.split(".")[-1]
            for mat, name in zip(v.chunk(3, dim=0), ("query", "key", "value")):
                out[f"{_AST}encoder.layer.{layer}.attention.attention.{name}.{kind}"] = mat
        else:
            out[k] = v
    return out


def from_hf_ast_state_dict(hf: Dict[str, torch.Tensor], n_freq: int, n_time: int) -> Dict[str, torch.Tensor]:
    """Inverse of to_hf_ast_state_dict.  The single position table is split as freq[f] = pos[f, 0] and
    time[t] = pos[0, t] - pos[0, 0]: exact whenever the table is additive (every converted MAEST model)."""
    hf = dict(hf)
    pos = hf.pop(_AST + "embeddings.position_embeddings")
    sd = {"new_pos_embed": pos[:, :2].clone()}
    grid = pos[0, 2:].reshape(n_freq, n_time, -1)
    sd["freq_new_pos_embed"] = grid[:, 0].t().reshape(1, -1, n_freq, 1).contiguous()
    sd["time_new_pos_embed"] = (grid[0] - grid[0, :1]).t().reshape(1, -1, 1, n_time).contiguous()
    layers = sorted({int(k.split(".")[3]) for k in hf if k.startswith(_AST + "encoder.layer.")})
    for l in layers:
        p = f"{_AST}encoder.layer.{l}.attention.attention."
        for kind in ("weight", "bias"):
            sd[f"blocks.{l}.attn.qkv.{kind}"] = torch.cat([hf.pop(p + f"{n}.{kind}") for n in ("query", "key", "value")])
    for k, v in hf.items():
        for a, b in reversed(_RENAMES):
            k = k.replace(b, a)
        sd[k] = v
    return sd
