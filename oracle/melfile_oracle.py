"""ORACLE (test infrastructure only -- never imported by the product): numpy restatement of the reference's
on-disk mel reader, pinned bit-exactly against the reference's own function by tests/golden/g8_melfile.npz
(written by oracle/gen_golden.py, which imports discogs/dataset.py from /root/reference).

  load_melspectrogram  <- DiscogsDataset.load_melspectrogram, discogs/dataset.py:69-140
  norm_func            <- DiscogsDataModule.get_norm_func,     discogs/datamodule.py:126-136
"""
import pathlib

import numpy as np


def load_melspectrogram(path, melspectrogram_size: int, n_bands: int = 96, offset: int = 0) -> np.ndarray:
    """-> float16 [1, n_bands, melspectrogram_size]; `offset` must be given (the random draw is the caller's)."""
    path = pathlib.Path(path)
    size = melspectrogram_size
    if path.suffix == ".npy":                                   # dataset.py:72-87
        mel = np.load(path).astype("float16")
        if mel.shape[0] < size:
            pad = size - mel.shape[0]
            mel = np.vstack([mel, np.zeros([pad, n_bands], dtype="float16")])
            mel = np.roll(mel, pad // 2, axis=0)
        else:
            mel = mel[:size, :]
    else:                                                       # dataset.py:88-132
        frames_num = path.stat().st_size // (2 * n_bands)
        skip_frames = max(offset + size - frames_num, 0)
        frames_to_read = size - skip_frames
        raw = np.fromfile(path, dtype="float16", count=frames_to_read * n_bands, offset=offset * n_bands * 2)
        mel = raw.reshape(frames_to_read, n_bands)
        if frames_to_read < size:
            pad = size - frames_to_read
            mel = np.vstack([mel, np.zeros([pad, n_bands], dtype="float16")])
            mel = np.roll(mel, pad // 2, axis=0)                # centre the padding
    return np.expand_dims(mel.T, 0)                             # dataset.py:135-137


def norm_func(x: np.ndarray, norm_mean: float = 2.06755686098554, norm_std: float = 1.268292820667291) -> np.ndarray:
    """datamodule.py:131 on a float16 array: numpy keeps float16 (both scalars are cast down first)."""
    return (x - norm_mean) / (norm_std * 2)
