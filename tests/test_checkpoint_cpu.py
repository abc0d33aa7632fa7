"""SURVEY 8f row 2 (checkpoint interop), CPU: maest_amd.checkpoint against the fixture captured from the imported
reference's checkpoint_filter_fn (oracle/gen_golden_checkpoint.py), the Lightning .ckpt loader, and the
HF-AST layout round trip checked against the oracle's token assembly."""
import os
import types

import numpy as np
import torch

from maest_amd import checkpoint as C
from oracle import maest_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g9_checkpoint.npz")


def synth(shape, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * 0.02)


def fake_model(grid):
    m = types.SimpleNamespace()
    m.num_tokens = 2
    m.patch_embed = types.SimpleNamespace(grid_size=grid, proj=types.SimpleNamespace(weight=torch.zeros(768, 1, 16, 16)))
    return m


def test_adapt_state_dict_matches_reference_fixture():
    g = np.load(GOLD)
    sd = {"pos_embed": synth((1, 2 + 24 * 24, 768), 1), "patch_embed.proj.weight": synth((768, 256), 2)}
    r = C.adapt_state_dict(sd, fake_model((9, 62)))
    assert "pos_embed" not in r and tuple(r["patch_embed.proj.weight"].shape) == tuple(g["deit_patch_shape"])
    assert np.array_equal(r["new_pos_embed"].numpy(), g["deit_new_pos_embed"])
    assert np.allclose(r["freq_new_pos_embed"].numpy()[0, :16, :, 0], g["deit_freq"], rtol=0, atol=1e-7)
    assert np.allclose(r["time_new_pos_embed"].numpy()[0, :16, 0, :], g["deit_time"], rtol=0, atol=1e-7)
    for name, grid in (("m30", (9, 187)), ("m5", (8, 31))):
        sd = {"new_pos_embed": synth((1, 2, 768), 3), "freq_new_pos_embed": synth((1, 768, 9, 1), 4),
              "time_new_pos_embed": synth((1, 768, 1, 62), 5)}
        r = C.adapt_state_dict(sd, fake_model(grid))
        assert r["freq_new_pos_embed"].shape == (1, 768, grid[0], 1) and r["time_new_pos_embed"].shape == (1, 768, 1, grid[1])
        assert np.allclose(r["freq_new_pos_embed"].numpy()[0, :16, :, 0], g[f"{name}_freq"], rtol=0, atol=1e-7)
        assert np.allclose(r["time_new_pos_embed"].numpy()[0, :16, 0, :], g[f"{name}_time"], rtol=0, atol=1e-7)
    # same grid: tables untouched
    r = C.adapt_state_dict(sd, fake_model((9, 62)))
    assert r["time_new_pos_embed"] is sd["time_new_pos_embed"]


def test_lightning_checkpoint_loader(tmp_path):
    from maest_amd import get_maest
    sd = O.make_state_dict(625, seed=5)
    ckpt = {"state_dict": {**{"net." + k: torch.zeros_like(v) for k, v in sd.items()},
                           **{"net_swa." + k: v for k, v in sd.items()}}}
    torch.save(ckpt, tmp_path / "last.ckpt")
    m = get_maest("discogs-maest-10s-fs-129e", pretrained=False, checkpoint=str(tmp_path / "last.ckpt"))
    assert torch.equal(m.blocks[3].mlp.fc1.weight, sd["blocks.3.mlp.fc1.weight"])          # SWA weights by default
    m30 = get_maest("discogs-maest-30s-pw-129e", pretrained=False)
    C.load_lightning_checkpoint(m30, tmp_path / "last.ckpt", discard_head=True, adapt=True)   # 10 s -> 30 s
    assert m30.time_new_pos_embed.shape[-1] == 187
    assert torch.equal(m30.blocks[0].attn.qkv.weight, sd["blocks.0.attn.qkv.weight"])


def test_hf_ast_layout_round_trip_and_token_order():
    sd = O.make_state_dict(625, seed=6)
    hf = C.to_hf_ast_state_dict(sd)
    assert not any("qkv" in k or k.startswith("head_dist") or "blocks." in k for k in hf)
    q = hf["audio_spectrogram_transformer.encoder.layer.4.attention.attention.key.weight"]
    assert torch.equal(q, sd["blocks.4.attn.qkv.weight"][768:1536])
    assert hf["classifier.dense.weight"].shape == (400, 768)
    assert "audio_spectrogram_transformer.encoder.layer.0.layernorm_before.weight" in hf
    assert "audio_spectrogram_transformer.layernorm.weight" in hf
    pos = hf["audio_spectrogram_transformer.embeddings.position_embeddings"]
    assert pos.shape == (1, 2 + 9 * 62, 768)
    # the recombined table, token by token, is what the hot path adds: token 2 + f * 62 + t gets freq[f] + time[t]
    x = torch.zeros(1, 768, 9, 62)
    toks = O.tokens_from_patches(x, sd)               # zeros + positional terms + cls/dist
    want = toks.clone()
    want[:, 0] -= sd["cls_token"][0, 0]
    want[:, 1] -= sd["dist_token"][0, 0]
    assert torch.allclose(pos, want, atol=1e-7)
    back = C.from_hf_ast_state_dict(hf, 9, 62)
    for k, v in sd.items():
        if k.startswith("head_dist."):
            continue
        if k in ("freq_new_pos_embed", "time_new_pos_embed"):
            continue                                  # only their sum is defined; checked below
        assert torch.equal(back[k], v), k
    assert torch.allclose(back["freq_new_pos_embed"] + back["time_new_pos_embed"],
                          sd["freq_new_pos_embed"] + sd["time_new_pos_embed"], atol=1e-7)
