"""The rows next to the training step against fixtures made by the
REAL reference (oracle/make_golden_frows.py -> tests/golden/frows.pt, ref_saver_ckpt/)."""
import os
import shutil
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, max_rel


@pytest.fixture(scope='module')
def frows():
    return load_golden('frows.pt')


# ---- f2: host restatement of the input pipeline ------------------------------------------------
def test_normalize_and_preemphasis_match_the_reference(frows):
    from segan_pytorch_amd.datasets import normalize_wave_minmax, pre_emphasize
    fx = frows['pcm']
    pcm = fx['pcm'].numpy()
    norm = normalize_wave_minmax(pcm)
    assert np.array_equal(np.asarray(norm, dtype=np.float64), fx['normalized'].numpy())
    pre = pre_emphasize(norm, fx['coef'])
    assert np.array_equal(np.asarray(pre, dtype=np.float64), fx['pre_emphasized'].numpy())


def test_host_deemphasis_matches_the_reference_loop(frows):
    from segan_pytorch_amd.datasets import de_emphasize
    fx = frows['deemph']
    got = de_emphasize(fx['y'].numpy(), fx['coef'])
    # the reference loop accumulates in float32 (numpy 2) — the filter in float64: 20x
    # amplification of 6e-8
    assert np.abs(got - fx['x'].numpy()).max() < 5e-6


# ---- f3: checkpoints written by the reference's Saver --------------------------------------------
def test_checkpoint_written_by_the_reference_saver_loads(frows, tmp_path):
    from segan_pytorch_amd.models import SEGAN
    from segan_pytorch_amd.models.core import Saver
    fx = frows['ckpt']
    assert fx['ours_loaded_by_reference'] is True        # asserted when the fixture was made
    dst = str(tmp_path / 'ck')
    shutil.copytree(os.path.join(GOLDEN, 'ref_saver_ckpt'), dst)
    torch.manual_seed(99)
    m = SEGAN(SimpleNamespace(**fx['opts']))
    # (1) through the Saver, like Model.load (core.py:176-186)
    saver = Saver(m.G, dst, max_ckpts=3, optimizer=None, prefix='EOE_G-')
    assert saver.load_weights() is True
    for k, v in fx['G_saved'].items():
        assert torch.equal(m.G.state_dict()[k], v), k
    # (2) as a pretrained checkpoint file (core.py:115-151), incl. the optimizer state it carries
    torch.manual_seed(98)
    m2 = SEGAN(SimpleNamespace(**fx['opts']))
    m2.G.load_pretrained(os.path.join(dst, 'weights_EOE_G-Generator-7.ckpt'), load_last=True)
    for k, v in fx['G_saved'].items():
        assert torch.equal(m2.G.state_dict()[k], v), k
    st = torch.load(os.path.join(dst, 'weights_EOE_G-Generator-7.ckpt'), weights_only=False)
    assert st['step'] == fx['step'] and 'optimizer' in st


# ---- GPU: f1 inference, f2 device prep, f4 SSNR ---------------------------------------------------
@pytest.mark.gpu
def test_generate_matches_the_reference(frows):
    """SEGAN.generate (chunked, batched here) and WSEGAN.generate (whole utterance) incl. the
    device de-emphasis against the reference's outputs; tolerance 2e-5 absolute on a [-1, 1]
    waveform (G forward 2e-6, de-emphasis recurrence x20)."""
    from segan_pytorch_amd.models import SEGAN, WSEGAN
    fx = frows['generate']
    m = SEGAN(SimpleNamespace(**fx['opts']))
    m.G.load_state_dict(fx['G0'])
    m = m.to('cuda')
    c_res, g_c = m.generate(fx['wav'].clone(), z=fx['z'].clone(), device='cuda')
    assert c_res.shape == tuple(fx['c_res'].shape) and c_res.dtype == np.float32
    assert np.abs(c_res - fx['c_res'].numpy()).max() < 2e-5
    assert max_rel(g_c, fx['g_c']) < 2e-5
    fw = frows['wgenerate']
    w = WSEGAN(SimpleNamespace(**fw['opts']))
    w.G.load_state_dict(fw['G0'])
    w = w.to('cuda')
    cw, _ = w.generate(fw['wav'].clone().to('cuda'), z=fw['z'].clone().to('cuda'))
    assert cw.shape == tuple(fw['c_res'].shape)
    assert np.abs(cw - fw['c_res'].numpy()).max() < 2e-5


@pytest.mark.gpu
def test_device_deemphasis_matches_the_reference_loop(frows):
    from segan_pytorch_amd import ops
    fx = frows['deemph']
    y = fx['y'].to('cuda')
    x = ops.de_emphasize(y, fx['coef'])
    assert (x.cpu() - fx['x']).abs().max().item() < 5e-6
    # rows are independent; lengths that are not a multiple of the slab
    y2 = torch.stack((fx['y'][:12345], fx['y'][20000:32345])).contiguous().to('cuda')
    x2 = ops.de_emphasize(y2, fx['coef'])
    assert (x2[0].cpu() - fx['x'][:12345]).abs().max().item() < 5e-6
    assert torch.equal(ops.de_emphasize(y, 0.0), y)


@pytest.mark.gpu
def test_device_pcm_prep_matches_the_reference_functions(frows):
    """segan_pcm16_prep against normalize_wave_minmax + pre_emphasize of the reference itself
    (se_dataset.py:108-117) on a slice in the middle of a wav and on its first slice."""
    from segan_pytorch_amd import ops
    fx = frows['pcm']
    pcm = fx['pcm']
    want = fx['pre_emphasized'].numpy().astype(np.float32)
    T = 16384
    rows, first, ref = [], [], []
    for beg in (0, 8192, 23616):
        if beg == 0:
            rows.append(torch.cat((pcm[:1], pcm[:T])))       # element 0 is ignored for a first slice
        else:
            rows.append(pcm[beg - 1:beg + T])
        first.append(1 if beg == 0 else 0)
        ref.append(want[beg:beg + T])
    block = torch.stack([torch.stack((r, r)) for r in rows]).contiguous().to('cuda')
    clean, noisy = ops.pcm16_prep(block, torch.tensor(first, dtype=torch.uint8, device='cuda'), fx['coef'])
    for i, w in enumerate(ref):
        assert np.array_equal(clean[i].cpu().numpy(), w), i
        assert np.array_equal(noisy[i].cpu().numpy(), w), i


@pytest.mark.gpu
def test_device_ssnr_matches_the_reference(frows):
    from segan_pytorch_amd import ops
    fx = frows['ssnr']
    snr, mean_seg, seg = ops.ssnr(fx['clean'].to('cuda'), fx['deg'].to('cuda'))
    assert seg.shape == tuple(fx['segmental'].shape)
    assert (seg.cpu().double() - fx['segmental']).abs().max().item() < 1e-4
    assert (snr.cpu().double() - fx['overall']).abs().max().item() < 1e-4
    assert (mean_seg.cpu().double() - fx['segmental'].mean(1)).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_train_with_validation_runs(frows, tmp_path):
    """SEGAN.train(va_dloader=...) validates on the GPU (SSNR objective) and keeps the best
    checkpoint, instead of raising."""
    from segan_pytorch_amd.models import SEGAN
    from segan_pytorch_amd.datasets import synthetic_pairs
    o = dict(frows['generate']['opts'])
    o.update(save_path=str(tmp_path), epoch=2, patience=5)
    torch.manual_seed(3)
    m = SEGAN(SimpleNamespace(**o)).to('cuda')
    c, n = synthetic_pairs(4, 1024, 1)
    loader = [[['u'] * 4, c, n, torch.zeros(4)]]
    vc, vn = synthetic_pairs(2, 16384, 2)
    va = [[['v'] * 2, vc, vn, torch.zeros(2)]]
    m.train(SimpleNamespace(**o), loader, None, o['l1_weight'], o['l1_dec_step'], o['l1_dec_epoch'],
            1000, va_dloader=va, device='cuda')
    names = os.listdir(str(tmp_path))
    assert any('best_Generator' in n_ for n_ in names), names
    ev, nev = m.evaluate(SimpleNamespace(**o), va, 1, do_noisy=True, device='cuda')
    assert len(ev['ssnr']) == 2 and len(nev['ssnr']) == 2 and all(-10 <= v <= 35 for v in ev['ssnr'])


# ---- the listening samples of the training loop (model.py:177-217) -------------------------------
def _check_train_samples(device, tmp_path):
    from scipy.io import wavfile
    from segan_pytorch_amd.models import SEGAN
    fx = load_golden('train_samples.pt')
    o = dict(fx['opts'])
    o['save_path'] = str(tmp_path)
    m = SEGAN(SimpleNamespace(**o))
    m.G.load_state_dict(fx['G0'])
    m.to(device)
    m.G.train()
    m.gen_train_samples(fx['clean'].to(device), fx['noisy'].to(device), fx['z'].to(device),
                        iteration=fx['iteration'])
    assert sorted(os.listdir(str(tmp_path))) == fx['files']
    for kind, pat in (('sample', 'sample_5-{}.wav'), ('gtruth', 'gtruth_{}.wav'),
                      ('noisy_wav', 'noisy_{}.wav'), ('dif', 'dif_{}.wav')):
        for i in range(3):
            rate, data = wavfile.read(str(tmp_path / pat.format(i)))
            assert rate == 16000 and data.dtype == np.float32
            # de-emphasis amplifies by up to 1/(1-0.95): compare relative to the signal's peak
            assert max_rel(torch.from_numpy(data), fx[kind][i]) < 2e-5, (kind, i)
    # a second call leaves the one-off files alone and adds the new samples
    m.gen_train_samples(fx['clean'].to(device), fx['noisy'].to(device), fx['z'].to(device),
                        iteration=9)
    assert len(os.listdir(str(tmp_path))) == len(fx['files']) + 3


def test_training_samples_host_logic_matches_reference(tmp_path):
    """gen_train_samples with the kernels emulated on CPU (tests/emu_ops.py): file set and
    contents against the reference's."""
    import emu_ops
    emu_ops.install()
    try:
        _check_train_samples('cpu', tmp_path)
    finally:
        emu_ops.uninstall()


@pytest.mark.gpu
def test_training_samples_match_reference_on_gpu(tmp_path):
    _check_train_samples('cuda', tmp_path)
