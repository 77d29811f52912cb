"""Batch format and waveform helpers of the reference's data side
(segan/datasets/se_dataset.py:21-29,108-126), plus the synthetic dataset the benchmark
and the tests use.  The on-disk dataset pipeline (wav slicing, caches) is outside the
accelerated path ("next"); `build_pcm_shard` / `PCMShardDataset` are
the MI355X-side replacement for it: slices are cut ONCE into an int16 shard file, batches
travel to the GPU as int16 (half the PCIe bytes of fp32) and are normalised and
pre-emphasised there by `segan_pcm16_prep`, bit-exactly as the reference does on the host.
"""
import json
import os

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


# ---- pre-sliced int16 shards ---------------------------------------
SHARD_MAGIC = 'segan-pcm16-shard-v1'


def build_pcm_shard(clean_dir, noisy_dir, out_prefix, slice_size=2 ** 14, stride=0.5,
                    max_samples=None):
    """Cut every clean/noisy wav pair into `slice_size` windows (same windows as
    `SEDataset` / se_dataset.py:62-88) and store them as raw int16 PCM:

      <out_prefix>.pcm16   int16 [n_items][2][slice_size+1]   (clean row, noisy row; element
                           0 = the wav sample preceding the window, needed by pre-emphasis)
      <out_prefix>.json    {magic, n_items, slice_size, names[], slice_idx[], first[]}

    The reference re-reads and re-normalises both full wav files for every item
    (se_dataset.py:190-198,309-353); at the HIP step's rate that starves the GPU."""
    import glob
    from scipy.io import wavfile
    clean_names = sorted(glob.glob(os.path.join(clean_dir, '*.wav')))
    noisy_names = sorted(glob.glob(os.path.join(noisy_dir, '*.wav')))
    if len(clean_names) != len(noisy_names) or len(clean_names) == 0:
        raise ValueError('No wav data found! Check your data path please')
    if max_samples is not None:
        clean_names, noisy_names = clean_names[:max_samples], noisy_names[:max_samples]
    rows, names, sidx, first = [], [], [], []
    for cpath, npath in zip(clean_names, noisy_names):
        c, n = wavfile.read(cpath)[1], wavfile.read(npath)[1]
        if c.dtype != np.int16 or n.dtype != np.int16:
            raise ValueError('pcm shards hold 16-bit PCM; {} is {}'.format(cpath, c.dtype))
        name = os.path.splitext(os.path.basename(cpath))[0]
        for si, (beg, end) in enumerate(slice_signal_index(min(len(c), len(n)), slice_size, stride)):
            item = np.empty((2, slice_size + 1), dtype=np.int16)
            for k, w in enumerate((c, n)):
                item[k, 1:] = w[beg:end]
                item[k, 0] = w[beg - 1] if beg > 0 else 0
            rows.append(item)
            names.append(name)
            sidx.append(si)
            first.append(1 if beg == 0 else 0)
    data = np.stack(rows) if rows else np.zeros((0, 2, slice_size + 1), np.int16)
    data.tofile(out_prefix + '.pcm16')
    with open(out_prefix + '.json', 'w') as f:
        json.dump({'magic': SHARD_MAGIC, 'n_items': len(rows), 'slice_size': slice_size,
                   'names': names, 'slice_idx': sidx, 'first': first}, f)
    return len(rows)


class PCMShardDataset(Dataset):
    """Items of a pcm16 shard: (uttname, int16 [2, T+1], first flag, slice_idx).  Use with
    `PCMShardCollate`, which turns a batch into the loader's [uttnames, clean, noisy,
    slice_idx] format with clean/noisy already on the GPU as fp32 — or, for the full-rate
    training loop, with `PCMShardLoader`, which gathers whole batches in worker processes."""

    def __init__(self, prefix):
        with open(prefix + '.json') as f:
            self.meta = json.load(f)
        if self.meta.get('magic') != SHARD_MAGIC:
            raise ValueError('{}.json is not a {} index'.format(prefix, SHARD_MAGIC))
        T = self.meta['slice_size']
        self.slice_size = T
        self.prefix = prefix
        self._data = None
        self._first = np.asarray(self.meta['first'], dtype=np.uint8)
        self._sidx = np.asarray(self.meta['slice_idx'], dtype=np.int64)

    @property
    def data(self):
        if self._data is None:      # opened lazily: every loader worker maps the file itself
            self._data = np.memmap(self.prefix + '.pcm16', dtype=np.int16, mode='r',
                                   shape=(self.meta['n_items'], 2, self.slice_size + 1))
        return self._data

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_data'] = None
        return d

    def __len__(self):
        return self.meta['n_items']

    def __getitem__(self, i):
        return (self.meta['names'][i], torch.from_numpy(np.array(self.data[i])),
                self.meta['first'][i], self.meta['slice_idx'][i])

    def gather(self, indices):
        """One batch as host tensors [names, int16 pcm [B, 2, T+1], uint8 first [B], slice_idx [B]]
        in ONE vectorised gather from the memory map (no per-item python work)."""
        idx = np.asarray(indices, dtype=np.int64)
        order = np.argsort(idx, kind='stable')          # ascending file offsets, then un-permute
        pcm = np.empty((len(idx), 2, self.slice_size + 1), dtype=np.int16)
        pcm[order] = self.data[idx[order]]
        names = self.meta['names']
        return [[names[i] for i in idx], torch.from_numpy(pcm), torch.from_numpy(self._first[idx]),
                torch.from_numpy(self._sidx[idx])]


class _BatchIndexDataset(Dataset):
    """Adapter: DataLoader(batch_size=None) hands a LIST of indices to __getitem__."""

    def __init__(self, shard):
        self.shard = shard

    def __len__(self):
        return len(self.shard)

    def __getitem__(self, indices):
        return self.shard.gather(indices)


class PCMShardLoader(object):
    """The full-rate input pipeline of `train.py --pcm_shard`: whole
    batches are gathered from the int16 shard by DataLoader WORKER processes (one vectorised
    gather each, prefetched two batches ahead, pinned by the loader's pin thread); the training
    process only issues the H2D copy of 20 MB of int16 and the `segan_pcm16_prep` kernel
    (normalisation + pre-emphasis on the GPU, bit-exact against the reference's host pipeline).
    Iterates [uttnames, clean[B,T], noisy[B,T], slice_idx[B]] with clean / noisy on the device,
    like the reference's loader plus its .to(device).  Same shuffling as DataLoader(shuffle=True)
    (a RandomSampler seeded from torch's global generator), or a DistributedSampler per rank."""

    def __init__(self, shard, batch_size, preemph, device, sampler=None, drop_last=False,
                 num_workers=2):
        from torch.utils.data import RandomSampler
        self.shard = shard
        self.sampler = sampler if sampler is not None else RandomSampler(shard)
        self.preemph = float(preemph)
        self.device = torch.device(device)
        self._side = None
        self._batch_size, self._drop_last, self._num_workers = batch_size, drop_last, num_workers
        self.loader = self._make_loader()
        self._sample_loader = None      # sample()'s own loader: see there

    def _make_loader(self, sampler=None, generator=None):
        from torch.utils.data import BatchSampler, DataLoader
        nw = self._num_workers
        return DataLoader(_BatchIndexDataset(self.shard), batch_size=None,
                          sampler=BatchSampler(sampler if sampler is not None else self.sampler,
                                               self._batch_size, self._drop_last),
                          num_workers=nw, pin_memory=True, prefetch_factor=2 if nw > 0 else None,
                          persistent_workers=nw > 0, generator=generator)

    def __len__(self):
        return len(self.loader)

    def _prep(self, item):
        from . import ops
        names, pcm, first, idx = item
        clean, noisy = ops.pcm16_prep(pcm.to(self.device, non_blocking=True),
                                      first.to(self.device, non_blocking=True), self.preemph)
        return [names, clean, noisy, idx]

    def _stage(self, item):
        """H2D copy + prep kernel of one batch on the loader's SIDE stream; (batch, event)."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._side):
            batch = self._prep(item)
            ev = torch.cuda.Event()
            ev.record(self._side)
        return batch, ev

    def __iter__(self):
        """Batch n+1 crosses PCIe (20 MB of int16 at batch 300: ~0.8 ms) and is normalised on a side
        stream while step n computes; the training stream only waits on the event of a copy that
        finished long ago."""
        if self.device.type != 'cuda':
            for item in self.loader:
                yield self._prep(item)
            return
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            (names, clean, noisy, idx), ev = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            clean.record_stream(main)
            noisy.record_stream(main)
            yield [names, clean, noisy, idx]

    def _ensure_sample_loader(self):
        if self._sample_loader is None:
            import copy
            from torch.utils.data import RandomSampler
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            if isinstance(self.sampler, RandomSampler):
                smp = RandomSampler(self.shard, replacement=self.sampler.replacement,
                                    num_samples=self.sampler._num_samples, generator=g)
            else:
                smp = copy.copy(self.sampler)
            self._sample_sampler, self._sample_gen = smp, g
            self._sample_loader = self._make_loader(smp, g)
        return self._sample_loader

    def sample(self):
        """One random batch per call from ONE live iterator, re-created only when the epoch is
        exhausted — for WSEGAN's `sample_dloader`, which the reference writes as
        `next(iter(dloader))` every step (model.py:526-535): on this loader that would reset the
        persistent workers, wait for an un-prefetched first batch (20 MB over IPC + pinning) and
        throw away up to three prefetched ones, every step (round-3 advice).  Batches then come
        from a shuffled pass without replacement instead of a fresh shuffle per step.

        The iterator lives on a SECOND DataLoader of the same shard (created on first use): a
        DataLoader with persistent workers has one shared `_iterator`, so sampling from the loader
        that `__iter__` walks — train / evaluate over this object while WSEGAN samples from it —
        would make the two iterators reset each other (dropped or duplicated batches; round-4
        advice).  That loader has its OWN sampler and its OWN torch.Generator (round-5 advice): a
        RandomSampler seeded by ONE draw from torch's global generator when the loader is created
        (a copy of a DistributedSampler keeps that sampler's seed and gets its own epoch counter), and
        the DataLoader's base seed comes from the same private generator — so after the first call
        `sample()` never touches torch's global CPU generator again and never changes the main
        sampler's epoch: the main loader's shuffles are what they would be without it, and the
        generator's z look-ahead (Generator.z_prefetch) stays valid across sample() calls, which is
        what lets WSEGAN.train keep the next z off the critical path (`sample_keeps_global_rng`).
        The next sample batch is staged (H2D + prep kernel) on the side stream while the caller
        computes on this one, like `__iter__` does."""
        self._ensure_sample_loader()

        def fetch():
            it = getattr(self, '_live', None)
            if it is None:
                it = self._live = iter(self._sample_loader)
            try:
                return next(it)
            except StopIteration:
                if hasattr(self._sample_sampler, 'set_epoch'):
                    self._epoch = getattr(self, '_epoch', 0) + 1
                    self._sample_sampler.set_epoch(self._epoch)
                it = self._live = iter(self._sample_loader)
                return next(it)

        if self.device.type != 'cuda':
            return self._prep(fetch())
        staged = getattr(self, '_sample_next', None)
        if staged is None:
            staged = self._stage(fetch())
        self._sample_next = self._stage(fetch())      # in flight under the caller's step
        (names, clean, noisy, idx), ev = staged
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        clean.record_stream(main)
        noisy.record_stream(main)
        return [names, clean, noisy, idx]

    # sample() draws from a private generator: WSEGAN.train may keep the z look-ahead on
    sample_keeps_global_rng = True

    def close(self):
        """Drop sample()'s loader (its worker processes and pinned buffers) and the staged batch."""
        self._sample_next = None
        self._live = None
        self._sample_loader = None


class PCMShardCollate(object):
    """collate_fn for `PCMShardDataset`: one pinned int16 staging copy, one H2D transfer,
    then normalisation + pre-emphasis on the device (`ops.pcm16_prep`)."""

    def __init__(self, preemph, device='cuda'):
        self.preemph = float(preemph)
        self.device = torch.device(device)

    def __call__(self, batch):
        from . import ops
        names = [b[0] for b in batch]
        pcm = torch.stack([b[1] for b in batch]).pin_memory()
        first = torch.tensor([b[2] for b in batch], dtype=torch.uint8).pin_memory()
        idx = torch.as_tensor([int(b[3]) for b in batch])
        clean, noisy = ops.pcm16_prep(pcm.to(self.device, non_blocking=True),
                                      first.to(self.device, non_blocking=True), self.preemph)
        return [names, clean, noisy, idx]
