#!/usr/bin/env python
"""Enhance wav files with a trained generator — the reference's clean.py entry point
(same flags; reads the `train.opts` written by train.py and a generator checkpoint) on
the HIP path.

    python clean.py --g_pretrained_ckpt ckpt/weights_EOE_G-Generator-N.ckpt \
        --cfg_file ckpt/train.opts --test_files noisy_dir --synthesis_path out --cuda
"""
import argparse
import glob
import json
import os
import random
import timeit
from types import SimpleNamespace

import numpy as np
import torch
from scipy.io import wavfile

from segan_pytorch_amd.datasets import normalize_wave_minmax, pre_emphasize
from segan_pytorch_amd.models import SEGAN, WSEGAN


def build_parser():
    p = argparse.ArgumentParser(description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('--g_pretrained_ckpt', type=str, default=None)
    p.add_argument('--test_files', type=str, nargs='+', default=None)
    p.add_argument('--h5', action='store_true', default=False)
    p.add_argument('--seed', type=int, default=111)
    p.add_argument('--synthesis_path', type=str, default='segan_samples')
    p.add_argument('--cuda', action='store_true', default=False)
    p.add_argument('--soundfile', action='store_true', default=False)
    p.add_argument('--cfg_file', type=str, default=None)
    return p


def main(opts):
    if opts.cfg_file is None or opts.test_files is None or opts.g_pretrained_ckpt is None:
        raise SystemExit('--cfg_file, --test_files and --g_pretrained_ckpt are required')
    if not (opts.cuda and torch.cuda.is_available()):
        raise SystemExit('segan_pytorch_amd runs only on an MI355X (HIP) device; pass --cuda on a '
                         'GPU machine (there is no CPU fallback)')
    if opts.h5:
        raise NotImplementedError('--h5 input is not implemented')
    with open(opts.cfg_file, 'r') as f:
        cfg = json.load(f)
    cfg.setdefault('reg_loss', 'l1_loss')      # older train.opts predate this flag
    args = SimpleNamespace(**cfg)
    args.cuda = True
    segan = (WSEGAN if getattr(args, 'wsegan', False) else SEGAN)(args)
    segan.G.load_pretrained(opts.g_pretrained_ckpt, True)
    segan.cuda()
    segan.G.eval()
    if len(opts.test_files) == 1 and os.path.isdir(opts.test_files[0]):
        twavs = sorted(glob.glob(os.path.join(opts.test_files[0], '*.wav')))
    else:
        twavs = opts.test_files
    print('Cleaning {} wavs'.format(len(twavs)))
    beg_t = timeit.default_timer()
    for t_i, twav in enumerate(twavs, start=1):
        rate, wav = wavfile.read(twav)
        wav = pre_emphasize(normalize_wave_minmax(wav), args.preemph)
        pwav = torch.as_tensor(wav, dtype=torch.float32).view(1, 1, -1).cuda()
        g_wav, _g_c = segan.generate(pwav, device='cuda')
        out_path = os.path.join(opts.synthesis_path, os.path.basename(twav))
        wavfile.write(out_path, int(16e3), np.asarray(g_wav, dtype=np.float32))
        end_t = timeit.default_timer()
        print('Cleaned {}/{}: {} in {} s'.format(t_i, len(twavs), twav, end_t - beg_t))
        beg_t = timeit.default_timer()


if __name__ == '__main__':
    opts = build_parser().parse_args()
    os.makedirs(opts.synthesis_path, exist_ok=True)
    random.seed(opts.seed)
    np.random.seed(opts.seed)
    torch.manual_seed(opts.seed)
    main(opts)
