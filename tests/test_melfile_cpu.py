"""SURVEY 8f row 1 (on-disk mel chunks -> input), CPU side: the oracle restatement against the fixture captured
from the imported reference reader (oracle/gen_golden_melfile.py), and the host logic of maest_amd.melfile."""
import os
import random

import numpy as np
import pytest

from oracle import melfile_oracle as MO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g8_melfile.npz")


def _frames(n, seed, bands):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.random((n, bands), dtype=np.float32) * 5.0).astype("float16")


def test_oracle_matches_reference_fixture(tmp_path):
    g = np.load(GOLD)
    size, bands = int(g["size"]), int(g["bands"])
    for name in g["names"]:
        name = str(name)
        n, seed, offset, suffix = int(g[f"{name}_frames"]), int(g[f"{name}_seed"]), int(g[f"{name}_offset"]), str(g[f"{name}_suffix"])
        fr = _frames(n, seed, bands)
        path = tmp_path / (name + suffix)
        if suffix == ".npy":
            np.save(path, fr)
        else:
            fr.tofile(path)
        raw = MO.load_melspectrogram(path, size, bands, offset)
        assert raw.dtype == np.float16 and raw.shape == (1, bands, size)
        assert np.array_equal(raw.view(np.uint16), g[f"{name}_raw"]), name
        assert np.array_equal(MO.norm_func(raw).view(np.uint16), g[f"{name}_norm"]), name


def test_reader_host_plan_matches_reference_logic(tmp_path):
    from maest_amd.melfile import MelFileReader
    rd = MelFileReader(tmp_path, clip_length=10)                 # 10 s -> 625 frames (dataset.py:52)
    assert rd.melspectrogram_size == 625
    for n in (1, 100, 625, 626, 3000):
        fr = _frames(n, n, 96)
        fr.tofile(tmp_path / f"f{n}.mel")
        # the random offset is drawn with the reference's call: random.randint(0, max(frames - size, 0))
        random.seed(n)
        expect = random.randint(0, max(n - 625, 0))
        random.seed(n)
        off, to_read = rd.plan(tmp_path / f"f{n}.mel")
        assert off == expect and to_read == min(n, 625)
        rows = rd.read_rows(tmp_path / f"f{n}.mel", off)
        assert rows.dtype == np.float16 and np.array_equal(rows, fr[off:off + to_read])
    # explicit offset running past the end: fewer frames are read (dataset.py:100-101)
    off, to_read = rd.plan(tmp_path / "f3000.mel", 2900)
    assert (off, to_read) == (2900, 100)
    with pytest.raises(ValueError):
        rd.read_rows(tmp_path / "f100.mel", 100)
    # .npy files are loaded whole and truncated (dataset.py:72-87)
    np.save(tmp_path / "g.npy", _frames(700, 7, 96))
    assert rd.read_rows(tmp_path / "g.npy").shape == (625, 96)


def test_reader_refuses_cpu_tensors(tmp_path):
    """no CPU fallback: assembling on a non-HIP device must fail loudly"""
    from maest_amd.melfile import MelFileReader
    from maest_amd._lib import MaestHipError
    rd = MelFileReader(tmp_path, clip_length=1, sample_rate=50, hop_size=1)
    _frames(60, 1, 96).tofile(tmp_path / "a.mel")
    with pytest.raises((MaestHipError, RuntimeError, AssertionError)):
        rd.load_batch(["a.mel"], "cpu", offsets=[0])


def test_exhaustive_plan_and_teacher_targets_match_the_reference_fixture(tmp_path):
    """tests/golden/g11_dataset_formats.npz (oracle/gen_golden_dataset.py: DiscogsDatasetExhaustive's chunk plan and
    DiscogsDatasetTS's hard teacher targets, captured from the imported reference) -- host logic, exact."""
    import os
    import numpy as np
    from maest_amd.melfile import MelFileReader, hard_teacher_target, load_teacher_targets
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_dataset_formats.npz"))
    bands = int(g["bands"])
    names = [f"a/track{i}.mmap" for i in range(len(g["frames"]))]
    os.makedirs(tmp_path / "a")
    for n, fr in zip(names, g["frames"]):
        np.zeros((int(fr), bands), "float16").tofile(tmp_path / n)          # only the sizes matter for the plan
    rd = MelFileReader(tmp_path, clip_length=int(g["clip_length"]), sample_rate=int(g["sample_rate"]),
                       hop_size=int(g["hop_size"]), n_bands=bands)
    for half in (0, 1):
        plan = rd.exhaustive_plan(names, half_overlapped=bool(half))
        assert [names.index(f) for f, _ in plan] == list(g[f"plan_half{half}_file"])
        assert [o for _, o in plan] == list(g[f"plan_half{half}_offset"])
        for f, o in plan:                                                    # every planned patch is readable
            assert 0 < rd.plan(tmp_path / f, int(o))[1] <= rd.melspectrogram_size
    assert rd.exhaustive_plan(["x.npy", "y.npy"]) == [("x.npy", 0), ("y.npy", 0)]
    thr = float(g["teacher_threshold"])
    want = g["teacher_hard"].view(np.float16)
    os.makedirs(tmp_path / "t" / "a")
    for n, lg in zip(names, g["teacher_logits"]):
        np.save(tmp_path / "t" / (n + ".logits.npy"), lg)
        got = hard_teacher_target(lg, thr)
        assert got.dtype == np.float16
    got = load_teacher_targets(names, tmp_path / "t", thr)
    assert np.array_equal(got.view(np.uint16), g["teacher_hard"])
    assert want[1].sum() == 1.0                                              # the arg-max fallback row
