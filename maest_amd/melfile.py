"""On-disk mel-spectrogram chunks -> device-resident network input (the step BEFORE the hot
path in real training).

Reference being replaced (paths relative to palonso/MAEST):
  * writer: helpers/melspectrogram_extractor.py:44-48 -- raw float16 rows ``[frames, 96]`` (``np.memmap``), no header;
  * reader: ``DiscogsDataset.load_melspectrogram`` discogs/dataset.py:69-140 -- frame count from the file size,
    random (or given) frame offset, ``frames_to_read = size - max(offset + size - frames, 0)``, zero padding
    centred by ``np.roll(pad // 2)``, transpose to ``[1, 96, T]``; ``.npy`` files are loaded whole and truncated;
  * ``norm_func`` discogs/datamodule.py:126-136 -- ``(x - mean) / (2 std)`` evaluated by numpy in float16.
In the reference 16 loader workers per GPU do this on the host, sample by sample.  Here the host only decides WHAT
to read (offsets: same ``random.randint`` call as the reference, so a shared seed reproduces its draws) and copies
the raw rows of a whole batch into one pinned buffer; padding, roll, transpose and normalisation run in one HIP
kernel (csrc/embed.hip: melfile_assemble_kernel) and the batch never exists on the host in its final form.
"""
from __future__ import annotations

import pathlib
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops


class MelFileReader:
    def __init__(self, base_dir=".", clip_length: int = 10, sample_rate: int = 16000, hop_size: int = 256,
                 n_bands: int = 96, norm_mean: float = 2.06755686098554, norm_std: float = 1.268292820667291):
        self.base_dir = pathlib.Path(base_dir)
        self.n_bands = n_bands
        self.melspectrogram_size = clip_length * sample_rate // hop_size       # dataset.py:52
        self.norm_mean = norm_mean
        self.norm_std = norm_std

    # ---- host side: which rows of which file (dataset.py:88-104) ---------------------------------------
    def plan(self, path, offset: Optional[int] = None) -> Tuple[int, int]:
        """-> (offset, frames_to_read) for a raw float16 file; draws the offset like the reference if None."""
        size = self.melspectrogram_size
        frames_num = pathlib.Path(path).stat().st_size // (2 * self.n_bands)
        if type(offset) is not int:
            max_frame = frames_num - size
            offset = random.randint(0, max(max_frame, 0))
        skip_frames = max(offset + size - frames_num, 0)
        return offset, size - skip_frames

    def read_rows(self, path, offset: Optional[int] = None) -> np.ndarray:
        """The raw float16 rows ``[frames_to_read <= size, n_bands]`` the reference would read from `path`."""
        path = pathlib.Path(path)
        size = self.melspectrogram_size
        if path.suffix == ".npy":                                            # dataset.py:72-87
            return np.ascontiguousarray(np.load(path).astype("float16")[:size, :])
        offset, frames_to_read = self.plan(path, offset)
        if frames_to_read <= 0:
            raise ValueError(f"{path}: offset {offset} is past the end of the file")
        fp = np.memmap(path, dtype="float16", mode="r", shape=(frames_to_read, self.n_bands),
                       offset=offset * self.n_bands * 2)
        rows = np.array(fp, dtype="float16")
        del fp
        return rows

    # ---- device side ---------------------------------------------------------------------------------
    def assemble(self, rows: Sequence[np.ndarray], device, normalize: bool = True) -> torch.Tensor:
        """list of raw row blocks -> fp32 ``[B, 1, n_bands, T]`` on `device` (one H2D copy + one kernel)."""
        counts = [int(r.shape[0]) for r in rows]
        if any(c <= 0 or c > self.melspectrogram_size for c in counts):
            raise ValueError("every clip needs between 1 and melspectrogram_size frames")
        total = sum(counts)
        dev = torch.device(device)
        pin = dev.type == "cuda"
        staging = torch.empty((total, self.n_bands), dtype=torch.float16, pin_memory=pin)
        starts = np.zeros(len(rows), dtype=np.int64)
        view = staging.numpy()
        o = 0
        for i, r in enumerate(rows):
            starts[i] = o
            view[o:o + counts[i]] = r
            o += counts[i]
        frames = staging.to(dev, non_blocking=True)
        row_start = torch.from_numpy(starts).to(dev, non_blocking=True)
        frames_read = torch.tensor(counts, dtype=torch.int32).to(dev, non_blocking=True)
        x = ops.melfile_assemble(frames, row_start, frames_read, self.melspectrogram_size, normalize,
                                 self.norm_mean, self.norm_std)
        return x.unsqueeze(1)

    def load_batch(self, filenames: Sequence, device, offsets: Optional[Sequence[Optional[int]]] = None,
                   normalize: bool = True) -> torch.Tensor:
        """``[DiscogsDataset[i][0] for i in batch]`` + norm_func, assembled on the device."""
        offsets = offsets if offsets is not None else [None] * len(filenames)
        rows = [self.read_rows(self.base_dir / f, o) for f, o in zip(filenames, offsets)]
        return self.assemble(rows, device, normalize)

    # ---- evaluation: every chunk of every track (DiscogsDatasetExhaustive.__init__, dataset.py:218-246) --------
    def exhaustive_plan(self, filenames: Sequence, half_overlapped: bool = False) -> List[Tuple[str, int]]:
        """(file, frame offset) pairs that tile each raw float16 track with patches of `melspectrogram_size` frames,
        hop = the patch (or half of it), keeping the patches that start within 110 % of the track (the tail is
        zero-padded by the reader); other file types get the single patch (file, 0)."""
        size = self.melspectrogram_size
        hop = size // 2 if half_overlapped else size
        names = [str(f) for f in filenames]
        if not names or pathlib.Path(names[0]).suffix != ".mmap":
            return [(f, 0) for f in names]
        plan = []
        for f in names:
            frames_num = (self.base_dir / f).stat().st_size // (2 * self.n_bands)
            if half_overlapped:
                frames_num -= hop
            n_patches = int((frames_num * 1.1) // hop)
            plan.extend((f, i * hop) for i in range(n_patches))
        return plan


def hard_teacher_target(logits: np.ndarray, threshold: float) -> np.ndarray:
    """``DiscogsDatasetTS.__getitem__`` (dataset.py:177-191): teacher logits as stored (``<file>.logits.npy``) ->
    float16 -> logistic -> {0,1} float16 at `threshold`; when no class passes, the arg-max class alone.  Host glue
    (a few hundred values per clip); scipy's ``expit`` as in the reference so that borderline classes fall the same way."""
    from scipy.special import expit
    t = expit(np.asarray(logits).astype("float16").squeeze())
    hard = (t > threshold).astype("float16")
    if not np.sum(hard):
        hard = np.zeros(hard.shape, dtype="float16")
        hard[np.argmax(t)] = 1.0
    return hard


def load_teacher_targets(filenames: Sequence, teacher_target_base_dir, threshold: float) -> np.ndarray:
    """float16 ``[B, C]`` hard teacher targets of a batch (one ``.logits.npy`` per clip, dataset.py:172-175)."""
    base = pathlib.Path(teacher_target_base_dir)
    return np.stack([hard_teacher_target(np.load(base / (str(f) + ".logits.npy")), threshold) for f in filenames])
