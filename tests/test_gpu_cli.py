"""End-to-end entry points on the GPU: train.py (synthetic data) writes reference-format
checkpoints + train.opts, clean.py enhances a wav with them."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy.io import wavfile

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_clean(tmp_path):
    ck = str(tmp_path / 'ckpt')
    cmd = [sys.executable, os.path.join(ROOT, 'train.py'), '--save_path', ck, '--synthetic', '8',
           '--batch_size', '4', '--epoch', '1', '--save_freq', '1', '--no_train_gen',
           '--genc_fmaps', '8', '16', '32', '--denc_fmaps', '8', '16', '32', '--genc_poolings',
           '4', '4', '4', '--denc_poolings', '4', '4', '4', '--z_dim', '32', '--slice_size', '1024',
           '--num_workers', '0']
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'btime' in out.stdout
    opts = json.load(open(os.path.join(ck, 'train.opts')))
    assert opts['bias'] is True and opts['batch_size'] == 4
    names = os.listdir(ck)
    g_ckpts = [n for n in names if n.startswith('weights_EOE_G-Generator-')]
    assert g_ckpts and 'EOE_G-checkpoints' in names and 'EOE_D-checkpoints' in names
    # a 1.3-chunk utterance of int16 noise -> clean.py
    wav_dir, out_dir = tmp_path / 'noisy', tmp_path / 'enh'
    wav_dir.mkdir()
    rng = np.random.default_rng(0)
    wavfile.write(str(wav_dir / 'a.wav'), 16000, (rng.standard_normal(21000) * 3000).astype(np.int16))
    cmd = [sys.executable, os.path.join(ROOT, 'clean.py'), '--g_pretrained_ckpt',
           os.path.join(ck, g_ckpts[0]), '--cfg_file', os.path.join(ck, 'train.opts'),
           '--test_files', str(wav_dir), '--synthesis_path', str(out_dir), '--cuda']
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rate, enh = wavfile.read(str(out_dir / 'a.wav'))
    assert rate == 16000 and enh.shape[0] == 21000 and np.isfinite(enh).all()


@pytest.mark.parametrize('extra,host', [([], False), (['--host_z'], True)])
def test_train_with_bf16_contractions(tmp_path, extra, host):
    """`train.py --precision bf16` (BASELINE config 5 through the entry point): the run trains with
    finite losses, the choice is recorded in train.opts, and z is drawn on the GPU by default —
    on the host, like the reference, with --host_z."""
    ck = str(tmp_path / 'ckpt')
    cmd = [sys.executable, os.path.join(ROOT, 'train.py'), '--save_path', ck, '--synthetic', '16',
           '--batch_size', '4', '--epoch', '1', '--save_freq', '1', '--no_train_gen', '--precision', 'bf16',
           '--genc_fmaps', '16', '32', '64', '--denc_fmaps', '16', '32', '64', '--genc_poolings',
           '4', '4', '4', '--denc_poolings', '4', '4', '4', '--z_dim', '64', '--slice_size', '1024',
           '--num_workers', '0'] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'btime' in out.stdout and 'nan' not in out.stdout.lower()
    opts = json.load(open(os.path.join(ck, 'train.opts')))
    assert opts['precision'] == 'bf16' and opts['host_z'] is host
    assert any(n.startswith('weights_EOE_G-Generator-') for n in os.listdir(ck))


def test_train_wsegan_snorm_from_a_pcm_shard(tmp_path):
    """The run_wsegan_train.sh flavour end to end: int16 shard -> GPU normalise / pre-emphasis
    -> WSEGAN step with the misaligned pair, spectral norm in D and Adam."""
    from segan_pytorch_amd.datasets import build_pcm_shard
    rng = np.random.default_rng(1)
    cd, nd = tmp_path / 'clean', tmp_path / 'noisy'
    cd.mkdir()
    nd.mkdir()
    for i in range(3):
        c = (rng.standard_normal(6000) * 4000).astype(np.int16)
        wavfile.write(str(cd / 'u{}.wav'.format(i)), 16000, c)
        wavfile.write(str(nd / 'u{}.wav'.format(i)), 16000,
                      (c + rng.standard_normal(6000) * 500).astype(np.int16))
    n = build_pcm_shard(str(cd), str(nd), str(tmp_path / 'sh'), slice_size=1024, stride=0.5)
    assert n >= 8
    ck = str(tmp_path / 'ckpt')
    cmd = [sys.executable, os.path.join(ROOT, 'train.py'), '--save_path', ck, '--pcm_shard',
           str(tmp_path / 'sh'), '--batch_size', '4', '--epoch', '1', '--save_freq', '1',
           '--wsegan', '--misalign_pair', '--dnorm_type', 'snorm', '--opt', 'adam',
           '--genc_fmaps', '8', '16', '32', '--denc_fmaps', '8', '16', '32', '--genc_poolings',
           '4', '4', '4', '--denc_poolings', '4', '4', '4', '--z_dim', '32', '--slice_size', '1024',
           '--num_workers', '0']
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    names = os.listdir(ck)
    assert any(n_.startswith('weights_EOE_G-Generator-') for n_ in names)
    # no --no_train_gen: the listening samples of model.py:744-747 are written at every log step
    assert any(n_.startswith('sample_') for n_ in names) and 'gtruth_0.wav' in names, names


def test_train_from_wav_dirs_with_a_validation_set(tmp_path):
    """train.py --clean_trainset/--noisy_trainset + --clean_valset/--noisy_valset (reference
    train.py:70-98): the per-epoch validation runs on the device (segmental SNR) and the best
    checkpoint is saved."""
    rng = np.random.default_rng(2)
    dirs = {}
    for split in ('tr', 'va'):
        cd, nd = tmp_path / (split + '_clean'), tmp_path / (split + '_noisy')
        cd.mkdir()
        nd.mkdir()
        for i in range(3):
            c = (rng.standard_normal(5000) * 4000).astype(np.int16)
            wavfile.write(str(cd / 'u{}.wav'.format(i)), 16000, c)
            wavfile.write(str(nd / 'u{}.wav'.format(i)), 16000,
                          (c + rng.standard_normal(5000) * 500).astype(np.int16))
        dirs[split] = (str(cd), str(nd))
    ck = str(tmp_path / 'ckpt')
    cmd = [sys.executable, os.path.join(ROOT, 'train.py'), '--save_path', ck,
           '--clean_trainset', dirs['tr'][0], '--noisy_trainset', dirs['tr'][1],
           '--clean_valset', dirs['va'][0], '--noisy_valset', dirs['va'][1],
           '--cache_dir', str(tmp_path / 'cache'),
           '--batch_size', '4', '--epoch', '2', '--save_freq', '50', '--no_train_gen',
           '--genc_fmaps', '8', '16', '32', '--denc_fmaps', '8', '16', '32', '--genc_poolings',
           '4', '4', '4', '--denc_poolings', '4', '4', '4', '--z_dim', '32', '--slice_size', '1024',
           '--num_workers', '0']
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
    assert 'Val obj (SSNR) improved' in out.stdout
    names = os.listdir(ck)
    assert any('best' in n_.lower() for n_ in names), names
