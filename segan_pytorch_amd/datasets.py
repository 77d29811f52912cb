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


def slice_signal_index(n_samples, window_size, stride):
    """(begin, end) pairs of the windows of a signal (se_dataset.py:62-88): hop =
    stride * window_size, last partial window dropped."""
    assert 0 < stride <= 1, stride
    assert window_size % 2 == 0, window_size
    hop = int(window_size * stride)
    return [(beg, beg + window_size) for beg in range(0, n_samples - window_size + 1, hop)]


class SEDataset(Dataset):
    """Noisy/clean wav-directory dataset in the reference's item format
    (se_dataset.py:128-371): every item is one `slice_size` window of a clean/noisy pair,
    int16 PCM -> [-1, 1] -> pre-emphasis, windows taken every `stride*slice_size` samples.

    Same constructor arguments as the reference.  Unlike it, the whole set is sliced once
    in memory at construction (no per-item wav re-read, no pickle cache): at the
    throughput of the HIP step a per-item file read would starve the GPU
    (SURVEY.md section 8f)."""

    def __init__(self, clean_dir, noisy_dir, preemph, cache_dir='.', split='train',
                 slice_size=2 ** 14, stride=0.5, max_samples=None, do_cache=False, verbose=False,
                 slice_workers=2, preemph_norm=False, random_scale=[1]):
        import glob
        import os
        from scipy.io import wavfile
        super().__init__()
        clean_names = sorted(glob.glob(os.path.join(clean_dir, '*.wav')))
        noisy_names = sorted(glob.glob(os.path.join(noisy_dir, '*.wav')))
        if len(clean_names) != len(noisy_names) or len(clean_names) == 0:
            raise ValueError('No wav data found! Check your data path please')
        if max_samples is not None:
            clean_names, noisy_names = clean_names[:max_samples], noisy_names[:max_samples]
        self.preemph, self.preemph_norm = preemph, preemph_norm
        self.slice_size, self.stride = slice_size, stride
        self.random_scale = list(random_scale)
        self.items = []
        for cpath, npath in zip(clean_names, noisy_names):
            c = self._read(wavfile.read(cpath)[1])
            n = self._read(wavfile.read(npath)[1])
            name = os.path.splitext(os.path.basename(cpath))[0]
            for si, (beg, end) in enumerate(slice_signal_index(min(len(c), len(n)), slice_size,
                                                               stride)):
                self.items.append((name, torch.from_numpy(np.ascontiguousarray(c[beg:end])).float(),
                                   torch.from_numpy(np.ascontiguousarray(n[beg:end])).float(), si))
        if verbose:
            print('SEDataset[{}]: {} slices from {} files'.format(split, len(self.items),
                                                                  len(clean_names)))

    def _read(self, wav):
        if self.preemph_norm:
            return normalize_wave_minmax(pre_emphasize(wav, self.preemph))
        return pre_emphasize(normalize_wave_minmax(wav), self.preemph)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        name, c, n, si = self.items[i]
        if len(self.random_scale) > 1 or self.random_scale[0] != 1:
            import random as _r
            s = _r.choice(self.random_scale)
            c, n = c * s, n * s
        return name, c, n, si
