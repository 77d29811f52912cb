"""Batch format and waveform helpers of the reference's data side
(segan/datasets/se_dataset.py:21-29,108-126), plus the synthetic dataset the benchmark
and the tests use.  The on-disk dataset pipeline (wav slicing, caches) is outside the
accelerated path ("next").
"""
import numpy as np
import torch
from torch.utils.data import Dataset


def pre_emphasize(x, coef=0.95):
    """y[0] = x[0]; y[n] = x[n] - coef*x[n-1]   (se_dataset.py:111-117)."""
    if coef <= 0:
        return x
    x = np.asarray(x)
    x0 = np.reshape(x[0], (1,))
    diff = x[1:] - coef * x[:-1]
    return np.concatenate((x0, diff), axis=0)


def de_emphasize(y, coef=0.95):
    """Inverse first-order IIR x[n] = coef*x[n-1] + y[n]   (se_dataset.py:119-126)."""
    if coef <= 0:
        return y
    y = np.asarray(y)
    x = np.zeros(y.shape[0], dtype=np.float32)
    x[0] = y[0]
    # scipy.signal.lfilter is the vectorised form of the reference's python loop
    from scipy.signal import lfilter
    x = lfilter([1.0], [1.0, -coef], y.astype(np.float64)).astype(np.float32)
    return x


def normalize_wave_minmax(x):
    """int16 PCM -> [-1, 1]   (se_dataset.py:108-109)."""
    return (2. / 65535.) * (np.asarray(x, dtype=np.float64) - 32767.) + 1.


def collate_fn(batch):
    """[(uttname, clean, noisy, slice_idx), ...] -> [uttnames, clean[B,T], noisy[B,T],
    slice_idx[B]]   (se_dataset.py:21-29)."""
    names = [b[0] for b in batch]
    clean = torch.stack([torch.as_tensor(b[1], dtype=torch.float32) for b in batch])
    noisy = torch.stack([torch.as_tensor(b[2], dtype=torch.float32) for b in batch])
    idx = torch.as_tensor([int(b[3]) for b in batch])
    return [names, clean, noisy, idx]


def synthetic_pairs(B, T=16384, seed=0, device='cpu'):
    """The synthetic noisy/clean pairs of SURVEY.md section 8(d): uniform 'clean' in
    [-1, 1) plus 0.1-sigma Gaussian noise, clamped."""
    g = torch.Generator().manual_seed(seed)
    clean = torch.rand(B, T, generator=g) * 2 - 1
    noisy = (clean + 0.1 * torch.randn(B, T, generator=g)).clamp(-1, 1)
    return clean.to(device), noisy.to(device)


class SyntheticSEDataset(Dataset):
    """Fixed-seed synthetic 16 kHz noisy/clean chunks in the loader's item format."""

    def __init__(self, n_items, slice_size=16384, seed=0):
        self.clean, self.noisy = synthetic_pairs(n_items, slice_size, seed)

    def __len__(self):
        return self.clean.shape[0]

    def __getitem__(self, i):
        return 'synthetic_{}'.format(i), self.clean[i], self.noisy[i], 0
