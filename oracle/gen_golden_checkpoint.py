#!/usr/bin/env python
"""Generate tests/golden/g9_checkpoint.npz from the IMPORTED reference (authoring container only).

TEST INFRASTRUCTURE.  Drives the reference's ``checkpoint_filter_fn`` (models/maest.py:1051-1118) with two
synthetic state dicts -- a DeiT-style one carrying ``pos_embed`` and a MAEST one with a 10 s time table loaded
into a 30 s model -- and stores probes of the adapted position tables; the inputs are regenerated from seeds.

    python oracle/gen_golden_checkpoint.py
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import import_reference, OUT  # noqa: E402


def synth(shape, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * 0.02)


def fake_model(grid):
    m = types.SimpleNamespace()
    m.num_tokens = 2
    m.patch_embed = types.SimpleNamespace(grid_size=grid, proj=types.SimpleNamespace(weight=torch.zeros(768, 1, 16, 16)))
    return m


def main():
    rm = import_reference()
    out = {}
    # 1) DeiT / ImageNet checkpoint: pos_embed [1, 2 + 24*24, 768] -> grid (9, 62)
    sd = {"pos_embed": synth((1, 2 + 24 * 24, 768), 1), "patch_embed.proj.weight": synth((768, 256), 2)}
    r = rm.checkpoint_filter_fn(sd, fake_model((9, 62)))
    out["deit_new_pos_embed"] = r["new_pos_embed"].numpy()
    out["deit_freq"] = r["freq_new_pos_embed"].numpy()[0, :16, :, 0]
    out["deit_time"] = r["time_new_pos_embed"].numpy()[0, :16, 0, :]
    out["deit_patch_shape"] = np.array(r["patch_embed.proj.weight"].shape)
    # 2) MAEST 10 s tables (9 x 62) loaded into a 30 s model (9 x 187) and into a 8 x 31 grid
    for name, grid in (("m30", (9, 187)), ("m5", (8, 31))):
        sd = {"new_pos_embed": synth((1, 2, 768), 3), "freq_new_pos_embed": synth((1, 768, 9, 1), 4),
              "time_new_pos_embed": synth((1, 768, 1, 62), 5)}
        r = rm.checkpoint_filter_fn(sd, fake_model(grid))
        out[f"{name}_freq"] = r["freq_new_pos_embed"].numpy()[0, :16, :, 0]
        out[f"{name}_time"] = r["time_new_pos_embed"].numpy()[0, :16, 0, :]
    np.savez_compressed(os.path.join(OUT, "g9_checkpoint.npz"), **out)
    print("wrote g9_checkpoint.npz", os.path.getsize(os.path.join(OUT, "g9_checkpoint.npz")), "B")


if __name__ == "__main__":
    main()
