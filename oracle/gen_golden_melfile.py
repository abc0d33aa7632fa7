#!/usr/bin/env python
"""Generate tests/golden/g8_melfile.npz from the IMPORTED reference reader (authoring container only).

TEST INFRASTRUCTURE.  Imports ``/root/reference/discogs/dataset.py`` (sacred stubbed as in gen_golden.py), writes
PCG64-seeded float16 mel files to a temp dir, calls the reference's ``DiscogsDataset.load_melspectrogram`` with
explicit offsets, applies the reference's normalisation expression (discogs/datamodule.py:131) to its float16
output, asserts that oracle/melfile_oracle.py reproduces both bit for bit, and stores inputs' recipe + outputs.

    python oracle/gen_golden_melfile.py
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import melfile_oracle as MO  # noqa: E402
from oracle.gen_golden import import_reference, REF, OUT  # noqa: E402  (sacred / timm / torchaudio stubs)

SIZE, BANDS = 50, 96
# (name, frames in the file, suffix, offset)
CASES = [("mid", 80, ".mel", 13), ("tail", 80, ".mel", 45), ("short_even", 30, ".mel", 0),
         ("short_odd", 31, ".mel", 0), ("exact", 50, ".mel", 0), ("npy_short", 30, ".npy", 0),
         ("npy_long", 80, ".npy", 0), ("one_frame", 1, ".mel", 0)]


def make_frames(n, seed):
    """log-mel-like values (0 .. ~5) in float16, the on-disk dtype"""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.random((n, BANDS), dtype=np.float32) * 5.0).astype("float16")


def main():
    import_reference()
    sys.path.insert(0, REF)
    import discogs.dataset as rd
    ds = object.__new__(rd.DiscogsDataset)          # the constructor wants a ground-truth pickle; the reader
    ds.melspectrogram_size = SIZE                   # itself only uses these two attributes
    ds.n_bands = BANDS
    out = {"size": np.int64(SIZE), "bands": np.int64(BANDS), "names": np.array([c[0] for c in CASES])}
    with tempfile.TemporaryDirectory() as td:
        for i, (name, n, suffix, offset) in enumerate(CASES):
            frames = make_frames(n, 100 + i)
            path = os.path.join(td, name + suffix)
            if suffix == ".npy":
                np.save(path, frames)
            else:
                frames.tofile(path)
            import pathlib
            ref = ds.load_melspectrogram(pathlib.Path(path), offset)
            assert ref.dtype == np.float16 and ref.shape == (1, BANDS, SIZE), (ref.dtype, ref.shape)
            mine = MO.load_melspectrogram(path, SIZE, BANDS, offset)
            assert np.array_equal(ref.view(np.uint16), mine.view(np.uint16)), name
            norm_mean, norm_std = 2.06755686098554, 1.268292820667291
            ref_n = (ref - norm_mean) / (norm_std * 2)          # discogs/datamodule.py:131, verbatim expression
            assert ref_n.dtype == np.float16
            mine_n = MO.norm_func(mine)
            assert np.array_equal(ref_n.view(np.uint16), mine_n.view(np.uint16)), name
            out[f"{name}_frames"] = np.int64(n)
            out[f"{name}_seed"] = np.int64(100 + i)
            out[f"{name}_offset"] = np.int64(offset)
            out[f"{name}_suffix"] = np.array(suffix)
            out[f"{name}_raw"] = ref.view(np.uint16)
            out[f"{name}_norm"] = ref_n.view(np.uint16)
            print(f"  [{name}] oracle == reference (raw and normalised), {n} frames, offset {offset}")
    np.savez_compressed(os.path.join(OUT, "g8_melfile.npz"), **out)
    print("wrote", os.path.join(OUT, "g8_melfile.npz"), os.path.getsize(os.path.join(OUT, "g8_melfile.npz")), "B")


if __name__ == "__main__":
    main()
