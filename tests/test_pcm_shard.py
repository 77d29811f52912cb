"""Pre-sliced int16 shards: the shard holds exactly the windows the
wav-directory dataset produces, and the GPU normalise + pre-emphasis kernel reproduces the
reference's host pipeline (se_dataset.py:108-117,196-197) bit for bit."""
import os

import numpy as np
import pytest
import torch

from segan_pytorch_amd.datasets import (PCMShardCollate, PCMShardDataset, SEDataset, build_pcm_shard,
                                        normalize_wave_minmax, pre_emphasize)


def _write_wavs(tmp_path, lens=(40000, 16384, 52011)):
    from scipy.io import wavfile
    rng = np.random.RandomState(0)
    cd, nd = tmp_path / 'clean', tmp_path / 'noisy'
    cd.mkdir()
    nd.mkdir()
    for i, n in enumerate(lens):
        c = rng.randint(-32768, 32768, size=n).astype(np.int16)
        c[0] = -32768 if i == 0 else c[0]           # extremes of the int16 range
        c[1] = 32767
        z = np.clip(c.astype(np.int32) + rng.randint(-3000, 3000, size=n), -32768, 32767).astype(np.int16)
        wavfile.write(str(cd / 'utt{}.wav'.format(i)), 16000, c)
        wavfile.write(str(nd / 'utt{}.wav'.format(i)), 16000, z)
    return str(cd), str(nd)


def test_shard_holds_the_dataset_windows(tmp_path):
    cd, nd = _write_wavs(tmp_path)
    n = build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ref = SEDataset(cd, nd, preemph=0.95, slice_size=16384, stride=0.5)
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    assert n == len(ds) == len(ref) > 4
    for i in range(len(ds)):
        name, pcm, first, si = ds[i]
        rname, rc, rn, rsi = ref[i]
        assert (name, si) == (rname, rsi)
        assert pcm.dtype == torch.int16 and tuple(pcm.shape) == (2, 16385)
        # host restatement of the device kernel on the stored row == the dataset's item
        for row, want in ((pcm[0], rc), (pcm[1], rn)):
            x = normalize_wave_minmax(row.numpy())
            y = x[1:] - 0.95 * x[:-1]
            if first:
                y[0] = x[1]
            assert np.array_equal(y.astype(np.float32), want.numpy())
    with pytest.raises(ValueError):
        open(str(tmp_path / 'bad.json'), 'w').write('{"magic": "x"}')
        PCMShardDataset(str(tmp_path / 'bad'))


@pytest.mark.gpu
@pytest.mark.parametrize('preemph', [0.95, 0.0])
def test_gpu_prep_is_bit_exact(tmp_path, preemph):
    cd, nd = _write_wavs(tmp_path)
    build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ref = SEDataset(cd, nd, preemph=preemph, slice_size=16384, stride=0.5)
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    names, clean, noisy, idx = PCMShardCollate(preemph, 'cuda')([ds[i] for i in range(len(ds))])
    assert clean.is_cuda and clean.dtype == torch.float32 and tuple(clean.shape) == (len(ds), 16384)
    for i in range(len(ds)):
        rname, rc, rn, rsi = ref[i]
        assert names[i] == rname and int(idx[i]) == rsi
        assert torch.equal(clean[i].cpu(), rc)
        assert torch.equal(noisy[i].cpu(), rn)


def test_batch_gather_equals_the_items(tmp_path):
    """PCMShardDataset.gather (the worker-side batch fetch of the full-rate loader): one
    vectorised gather returns exactly the stacked items, for unsorted and repeated indices."""
    cd, nd = _write_wavs(tmp_path)
    build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    idx = [5, 0, 3, 3, len(ds) - 1, 1]
    names, pcm, first, sidx = ds.gather(idx)
    assert pcm.dtype == torch.int16 and tuple(pcm.shape) == (len(idx), 2, 16385)
    for k, i in enumerate(idx):
        name, item, f, si = ds[i]
        assert names[k] == name and int(first[k]) == f and int(sidx[k]) == si
        assert torch.equal(pcm[k], item)
    import pickle
    ds2 = pickle.loads(pickle.dumps(ds))        # loader workers receive a pickled copy
    assert torch.equal(ds2.gather(idx)[1], pcm)


@pytest.mark.gpu
def test_full_rate_loader_matches_the_reference_pipeline(tmp_path):
    """PCMShardLoader (worker-process gathers, pinned prefetch, GPU prep) yields, over one epoch,
    exactly the items of the reference's host pipeline — every item once, bit for bit."""
    from segan_pytorch_amd.datasets import PCMShardLoader
    cd, nd = _write_wavs(tmp_path)
    build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ref = SEDataset(cd, nd, preemph=0.95, slice_size=16384, stride=0.5)
    want = {(ref[i][0], ref[i][3]): (ref[i][1], ref[i][2]) for i in range(len(ref))}
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    torch.manual_seed(3)
    loader = PCMShardLoader(ds, 4, 0.95, 'cuda', num_workers=2)
    seen = set()
    for _ in range(2):                              # two epochs through persistent workers
        n = 0
        for names, clean, noisy, idx in loader:
            assert clean.is_cuda and clean.dtype == torch.float32
            for k, name in enumerate(names):
                rc, rn = want[(name, int(idx[k]))]
                assert torch.equal(clean[k].cpu(), rc) and torch.equal(noisy[k].cpu(), rn)
                seen.add((name, int(idx[k])))
                n += 1
        assert n == len(ds)
    assert seen == set(want)


@pytest.mark.gpu
def test_loader_sample_keeps_one_live_iterator(tmp_path):
    """PCMShardLoader.sample() (WSEGAN's per-step sample_dloader): 2.5 epochs of calls draw from
    ONE iterator per epoch — iter() of the underlying DataLoader is entered three times, not once
    per call — and every epoch yields every item once, bit for bit."""
    from segan_pytorch_amd.datasets import PCMShardLoader
    cd, nd = _write_wavs(tmp_path)
    build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ref = SEDataset(cd, nd, preemph=0.95, slice_size=16384, stride=0.5)
    want = {(ref[i][0], ref[i][3]): (ref[i][1], ref[i][2]) for i in range(len(ref))}
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    loader = PCMShardLoader(ds, 4, 0.95, 'cuda', num_workers=0)      # in-process: the iterator logic is the same, no worker start-up
    entered = []
    loader._ensure_sample_loader()       # sample()'s own DataLoader (lazy otherwise)
    assert loader._sample_loader is not loader.loader
    real_iter = type(loader._sample_loader).__iter__

    class Counting(type(loader._sample_loader)):
        def __iter__(self):
            entered.append(1)
            return real_iter(self)
    loader._sample_loader.__class__ = Counting
    per_epoch = len(loader)
    seen = []
    calls = 2 * per_epoch + per_epoch // 2 + 1
    for _ in range(calls):
        names, clean, noisy, idx = loader.sample()
        for k, name in enumerate(names):
            rc, rn = want[(name, int(idx[k]))]
            assert torch.equal(clean[k].cpu(), rc) and torch.equal(noisy[k].cpu(), rn)
            seen.append((name, int(idx[k])))
    # one batch is staged ahead of the one handed out: calls + 1 fetches
    assert len(entered) == -(-(calls + 1) // per_epoch), (len(entered), calls, per_epoch)
    assert sorted(seen[:len(ds)]) == sorted(want) and sorted(seen[len(ds):2 * len(ds)]) == sorted(want)


@pytest.mark.gpu
def test_loader_sample_does_not_disturb_an_epoch_in_progress(tmp_path):
    """Round-4 advice: iterating the loader (train / evaluate) while WSEGAN samples from the same
    object.  With one shared DataLoader the two iterators reset each other; with sample() on its
    own DataLoader an epoch interleaved with sample() calls still yields every item exactly once,
    and so does the sample stream."""
    from segan_pytorch_amd.datasets import PCMShardLoader
    cd, nd = _write_wavs(tmp_path)
    build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    loader = PCMShardLoader(ds, 4, 0.95, 'cuda', num_workers=1)
    epoch, sampled = [], []
    for names, clean, noisy, idx in loader:
        epoch += [(n, int(i)) for n, i in zip(names, idx)]
        sn, sc, sno, si = loader.sample()
        sampled += [(n, int(i)) for n, i in zip(sn, si)]
        assert torch.isfinite(sc).all() and torch.isfinite(clean).all()
    assert len(epoch) == len(ds) == len(set(epoch))
    assert len(sampled) == len(ds) == len(set(sampled))


@pytest.mark.gpu
def test_loader_sample_leaves_the_global_generator_alone(tmp_path):
    """Round-5 advice: after its first call sample() draws from its own sampler and generator —
    torch's global CPU generator stands where it stood across 2.5 sample epochs (the RandomSampler
    reseeds and the DataLoader base seed at every wrap-around used to take from it, which broke the
    generator's z look-ahead), and the main sampler's next shuffle is what it would have been."""
    from segan_pytorch_amd.datasets import PCMShardLoader
    cd, nd = _write_wavs(tmp_path)
    build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    loader = PCMShardLoader(ds, 4, 0.95, 'cuda', num_workers=0)
    assert loader.sample_keeps_global_rng
    loader.sample()                                  # creates the loader: ONE global draw, here
    torch.manual_seed(5)
    want = [list(b) for b in torch.utils.data.BatchSampler(torch.utils.data.RandomSampler(ds), 4, False)]
    torch.manual_seed(5)
    state = torch.get_rng_state()
    for _ in range(2 * len(loader) + len(loader) // 2 + 1):
        loader.sample()
    assert torch.equal(torch.get_rng_state(), state)
    got = [list(b) for b in loader.loader.sampler]   # the MAIN loader's next epoch order
    assert got == want
    loader.close()
    assert loader._sample_loader is None


@pytest.mark.gpu
def test_wsegan_train_on_the_shard_loader_with_and_without_z_lookahead(tmp_path):
    """WSEGAN.train over PCMShardLoader.sample() — the full-rate input path of train.py --wsegan
    --pcm_shard: the next step's z is drawn one step ahead on a host thread (round 6), which is only
    the reference's stream because sample() draws from a private generator.  Same seeds with the
    look-ahead on (the default) and off (opts.prefetch_z False): the same weights, bit for bit."""
    import random
    from types import SimpleNamespace
    from conftest import load_golden
    from segan_pytorch_amd import ops
    from segan_pytorch_amd.datasets import PCMShardLoader
    from segan_pytorch_amd.models import WSEGAN
    cd, nd = _write_wavs(tmp_path)
    build_pcm_shard(cd, nd, str(tmp_path / 'sh'), slice_size=16384, stride=0.5)
    ds = PCMShardDataset(str(tmp_path / 'sh'))
    o = dict(load_golden('tiny_wsegan2.pt')['opts'])
    o.update(save_path=str(tmp_path), epoch=2, dpool_slen=256, slice_size=16384, batch_size=4)
    old = ops.get_deterministic()
    ops.set_deterministic(True)
    try:
        out = {}
        for pf in (True, False):
            random.seed(3)
            np.random.seed(3)
            torch.manual_seed(3)
            m = WSEGAN(SimpleNamespace(**o)).to('cuda')
            loader = PCMShardLoader(ds, 4, 0.95, 'cuda', num_workers=0)
            seen = []
            real = m.wgan_step
            m.wgan_step = lambda *a, **k: (seen.append(m.G.z_prefetch), real(*a, **k))[1]
            oo = dict(o, prefetch_z=pf)
            m.train(SimpleNamespace(**oo), loader, None, o['l1_weight'], o['l1_dec_step'], o['l1_dec_epoch'],
                    1000, va_dloader=None, device='cuda')
            n = 2 * len(loader)
            assert seen == ([True] * (n - 1) + [False] if pf else [False] * n), (pf, seen)
            out[pf] = {k: v.detach().cpu().clone() for k, v in list(m.G.state_dict().items()) +
                       [('D.' + k, v) for k, v in m.D.state_dict().items()]}
            loader.close()
        for k, v in out[True].items():
            assert torch.equal(v, out[False][k]), k
    finally:
        ops.set_deterministic(old)
