#!/usr/bin/env python
"""Generate tests/golden/g11_dataset_formats.npz from the IMPORTED reference dataset classes (authoring container only).

TEST INFRASTRUCTURE.  Two host-side formats either side of the mel reader:
  * the exhaustive chunk plan of ``DiscogsDatasetExhaustive.__init__`` (discogs/dataset.py:196-246): (file, offset)
    pairs covering each track with a 10 % zero-pad margin, optionally half-overlapped;
  * the hard teacher targets of ``DiscogsDatasetTS.__getitem__`` (:168-193): ``<file>.logits.npy`` -> float16 ->
    expit -> threshold -> float16 {0,1}, the arg-max class when nothing passes.
The reference objects are built on temp files (PCG64-seeded), their outputs stored next to the recipe of the inputs.

    python oracle/gen_golden_dataset.py
"""
import os
import pathlib
import pickle
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import import_reference, REF, OUT  # noqa: E402

BANDS, CLIP, SR, HOP = 96, 2, 16000, 256                # 2 s clips: 125 frames per patch
FRAMES = [125, 124, 126, 300, 1, 63, 62, 700, 137, 250]  # frames in each synthetic track
N_CLASSES = 12


def main():
    import_reference()
    sys.path.insert(0, REF)
    import discogs.dataset as rd
    out = {"frames": np.array(FRAMES, np.int64), "bands": np.int64(BANDS), "clip_length": np.int64(CLIP),
           "sample_rate": np.int64(SR), "hop_size": np.int64(HOP)}
    with tempfile.TemporaryDirectory() as td:
        names = [f"a/track{i}.mmap" for i in range(len(FRAMES))]
        os.makedirs(os.path.join(td, "a"))
        rng = np.random.Generator(np.random.PCG64(5))
        gt = {}
        for n, fr in zip(names, FRAMES):
            (rng.random((fr, BANDS), dtype=np.float32) * 5).astype("float16").tofile(os.path.join(td, n))
            gt[n] = rng.random(N_CLASSES) < 0.3
        gtf = os.path.join(td, "gt.pk")
        pickle.dump(gt, open(gtf, "wb"))
        for half in (False, True):
            ds = rd.DiscogsDatasetExhaustive(gtf, td, SR, CLIP, HOP, BANDS, half)
            plan = [ds.filenames_with_patch[i] for i in range(len(ds))]
            out[f"plan_half{int(half)}_file"] = np.array([names.index(str(f)) for f, _ in plan], np.int64)
            out[f"plan_half{int(half)}_offset"] = np.array([o for _, o in plan], np.int64)
            print(f"  exhaustive plan (half_overlapped={half}): {len(plan)} patches")
        # teacher targets: logits with a spread that leaves some rows empty at the threshold, some borderline
        tdir = os.path.join(td, "teacher")
        os.makedirs(os.path.join(tdir, "a"))
        thr = 0.45
        logits = rng.standard_normal((len(names), 1, N_CLASSES)).astype(np.float32) * 1.5
        logits[1] = -3.0 - rng.random((1, N_CLASSES))            # nothing passes: arg-max fallback
        logits[2, 0, :4] = np.log(thr / (1 - thr)) + np.array([-2e-3, -1e-4, 1e-4, 2e-3], np.float32)   # around the threshold
        for n, lg in zip(names, logits):
            np.save(os.path.join(tdir, n + ".logits.npy"), lg)
        ts = object.__new__(rd.DiscogsDatasetTS)
        ts.filenames = dict(enumerate(names)); ts.groundtruth = gt; ts.base_dir = td
        ts.melspectrogram_size = CLIP * SR // HOP; ts.n_bands = BANDS
        ts.teacher_target_base_dir = tdir; ts.teacher_target_threshold = thr
        hard, tgt = [], []
        for i in range(len(names)):
            mel, fn, target, h = ts[i]
            assert h.dtype == np.float16 and target.dtype == np.float16 and fn == names[i]
            hard.append(h); tgt.append(target)
        out["teacher_logits"] = logits
        out["teacher_threshold"] = np.float64(thr)
        out["teacher_hard"] = np.stack(hard).view(np.uint16)
        out["targets"] = np.stack(tgt).view(np.uint16)
        out["groundtruth"] = np.stack([gt[n] for n in names])
        print(f"  teacher targets: {int(np.stack(hard).sum())} active of {np.stack(hard).size}")
    np.savez_compressed(os.path.join(OUT, "g11_dataset_formats.npz"), **out)
    print("wrote g11_dataset_formats.npz")


if __name__ == "__main__":
    main()
