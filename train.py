#!/usr/bin/env python
"""Train SEGAN+ / WSEGAN on MI355X — the reference's train.py entry point (same flags,
same defaults, same `train.opts` dump) driving the HIP engine in `segan_pytorch_amd`.

    python train.py --save_path ckpt_segan+ --clean_trainset ... --noisy_trainset ... \
        --batch_size 300 --no_train_gen
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...

Additions over the reference: `--synthetic N` trains on N fixed-seed synthetic chunk
pairs (no dataset needed), `--pcm_shard PREFIX` trains from a pre-sliced int16 shard whose
batches are normalised / pre-emphasised on the GPU, and under torch.distributed.run every
rank trains on its shard of each epoch with RCCL gradient averaging.
"""
import argparse
import json
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader

from segan_pytorch_amd import distributed as sdist
from segan_pytorch_amd import losses
from segan_pytorch_amd.datasets import (PCMShardCollate, PCMShardDataset, SEDataset,
                                        SyntheticSEDataset, collate_fn)
from segan_pytorch_amd.models import SEGAN, WSEGAN

_FMAPS = [64, 128, 256, 512, 1024]
_POOLS = [4, 4, 4, 4, 4]

# (flag, kwargs) — names and defaults are the reference's (train.py:101-247)
FLAGS = [
    ('--save_path', dict(type=str, default='seganv1_ckpt')),
    ('--d_pretrained_ckpt', dict(type=str, default=None)),
    ('--g_pretrained_ckpt', dict(type=str, default=None)),
    ('--cache_dir', dict(type=str, default='data_cache')),
    ('--clean_trainset', dict(type=str, default='data/clean_trainset')),
    ('--noisy_trainset', dict(type=str, default='data/noisy_trainset')),
    ('--clean_valset', dict(type=str, default=None)),
    ('--noisy_valset', dict(type=str, default=None)),
    ('--h5_data_root', dict(type=str, default=None)),
    ('--h5', dict(action='store_true', default=False)),
    ('--data_stride', dict(type=float, default=0.5)),
    ('--seed', dict(type=int, default=111)),
    ('--epoch', dict(type=int, default=100)),
    ('--patience', dict(type=int, default=100)),
    ('--batch_size', dict(type=int, default=100)),
    ('--save_freq', dict(type=int, default=50)),
    ('--slice_size', dict(type=int, default=16384)),
    ('--opt', dict(type=str, default='rmsprop')),
    ('--l1_dec_epoch', dict(type=int, default=100)),
    ('--l1_weight', dict(type=float, default=100)),
    ('--l1_dec_step', dict(type=float, default=1e-5)),
    ('--g_lr', dict(type=float, default=0.00005)),
    ('--d_lr', dict(type=float, default=0.00005)),
    ('--preemph', dict(type=float, default=0.95)),
    ('--max_samples', dict(type=int, default=None)),
    ('--eval_workers', dict(type=int, default=2)),
    ('--slice_workers', dict(type=int, default=1)),
    ('--num_workers', dict(type=int, default=1)),
    ('--no-cuda', dict(action='store_true', default=False)),
    ('--random_scale', dict(type=float, nargs='+', default=[1])),
    ('--no_train_gen', dict(action='store_true', default=False)),
    ('--preemph_norm', dict(action='store_true', default=False)),
    ('--wsegan', dict(action='store_true', default=False)),
    ('--aewsegan', dict(action='store_true', default=False)),
    ('--vanilla_gan', dict(action='store_true', default=False)),
    ('--no_bias', dict(action='store_true', default=False)),
    ('--n_fft', dict(type=int, default=2048)),
    ('--reg_loss', dict(type=str, default='l1_loss')),
    ('--skip_merge', dict(type=str, default='concat')),
    ('--skip_type', dict(type=str, default='alpha')),
    ('--skip_init', dict(type=str, default='one')),
    ('--skip_kwidth', dict(type=int, default=11)),
    ('--gkwidth', dict(type=int, default=31)),
    ('--genc_fmaps', dict(type=int, nargs='+', default=list(_FMAPS))),
    ('--genc_poolings', dict(type=int, nargs='+', default=list(_POOLS))),
    ('--z_dim', dict(type=int, default=1024)),
    ('--gdec_fmaps', dict(type=int, nargs='+', default=None)),
    ('--gdec_poolings', dict(type=int, nargs='+', default=None)),
    ('--gdec_kwidth', dict(type=int, default=None)),
    ('--gnorm_type', dict(type=str, default=None)),
    ('--no_z', dict(action='store_true', default=False)),
    ('--no_skip', dict(action='store_true', default=False)),
    ('--pow_weight', dict(type=float, default=0.001)),
    ('--misalign_pair', dict(action='store_true', default=False)),
    ('--interf_pair', dict(action='store_true', default=False)),
    ('--denc_fmaps', dict(type=int, nargs='+', default=list(_FMAPS))),
    ('--dpool_type', dict(type=str, default='none')),
    ('--dpool_slen', dict(type=int, default=16)),
    ('--dkwidth', dict(type=int, default=None)),
    ('--denc_poolings', dict(type=int, nargs='+', default=list(_POOLS))),
    ('--dnorm_type', dict(type=str, default='bnorm')),
    ('--phase_shift', dict(type=int, default=5)),
    ('--sinc_conv', dict(action='store_true', default=False)),
    # ---- additions ----
    ('--synthetic', dict(type=int, default=0,
                         help='train on this many fixed-seed synthetic chunk pairs')),
    ('--sync_bn', dict(action='store_true', default=False,
                       help='data parallel: take D\'s BatchNorm statistics over the global batch '
                            '(N ranks of batch b behave like one process at batch N*b)')),
    ('--device_z', dict(action='store_true', default=False,
                        help='draw the generator\'s z on the GPU (per-rank generator) instead of on the '
                             'host like the reference (generator.py:197): no host randn + copy per step, '
                             'but not the reference\'s RNG stream')),
    ('--precision', dict(type=str, default=None, choices=['fp32', 'bf16x3', 'bf16'],
                         help='contraction precision of the conv / deconv kernels (default: $SEGAN_PRECISION '
                              'or fp32 = exact fp32 MFMA).  bf16 = mixed precision on the bf16 matrix cores '
                              '(BASELINE config 5: bf16 operands, fp32 accumulation, everything else fp32; '
                              'tolerances in DESIGN.md section 6); bf16x3 = fp32 operands split exactly into '
                              '3 bf16 planes (fp32-class results).  With bf16 z is drawn on the GPU by default '
                              '(--device_z): a step takes about as long as the single-threaded host randn of '
                              'one z, so the host draw would be the critical path; --host_z keeps it')),
    ('--host_z', dict(action='store_true', default=False,
                      help='with --precision bf16: draw z on the host like the reference anyway')),
    ('--no_prefetch_z', dict(action='store_true', default=False,
                             help='draw every z inside its own step instead of one step ahead on a host '
                                  'thread (same numbers either way as long as nothing else takes from torch\'s global CPU '
                                  'generator inside an epoch; if something does — a num_workers=0 dataset using '
                                  'torch RNG, a hook — it is detected and the look-ahead switches itself off '
                                  'with a warning: Generator._host_z)')),
    ('--deterministic', dict(action='store_true', default=False,
                             help='bit-reproducible kernels: the weight-gradient and dense-head contraction '
                                  'splits are added in a fixed order instead of with fp32 atomics '
                                  '(1-2 %% slower)')),
    ('--pcm_shard', dict(type=str, default=None,
                         help='prefix of a pre-sliced int16 shard (scripts/make_pcm_shard.py): batches '
                              'are normalised and pre-emphasised on the GPU')),
]


def build_parser():
    p = argparse.ArgumentParser(description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    for flag, kw in FLAGS:
        p.add_argument(flag, **kw)
    return p


def main(opts):
    if getattr(opts, 'sync_bn', False):
        os.environ['SEGAN_SYNC_BN'] = '1'
    rank, world, local = sdist.init_from_env()
    use_cuda = torch.cuda.is_available() and not opts.no_cuda
    if not use_cuda:
        raise SystemExit('segan_pytorch_amd trains only on an MI355X (HIP) device: no GPU is '
                         'visible (and there is no CPU fallback)')
    device = torch.device('cuda', local if world > 1 else 0)
    torch.cuda.set_device(device)
    opts.cuda = True
    random.seed(opts.seed)
    np.random.seed(opts.seed)
    torch.manual_seed(opts.seed)
    torch.cuda.manual_seed_all(opts.seed)
    if opts.aewsegan:
        raise NotImplementedError('AEWSEGAN is broken in the reference (model.py:823) and is '
                                  'not implemented')
    segan = (WSEGAN if opts.wsegan else SEGAN)(opts)
    from segan_pytorch_amd import ops as _sops
    if getattr(opts, 'precision', None):
        _sops.set_precision(opts.precision)
    if _sops.get_precision() == 'bf16' and not getattr(opts, 'host_z', False):
        opts.device_z = True
    if getattr(opts, 'device_z', False):
        segan.G.z_generator = torch.Generator(device=device).manual_seed(opts.seed + rank)
    opts.prefetch_z = not getattr(opts, 'no_prefetch_z', False)
    if getattr(opts, 'deterministic', False):
        from segan_pytorch_amd import ops as _ops
        _ops.set_deterministic(True)
    segan.to(device)
    print('Total model parameters: ', segan.get_n_params())
    if opts.g_pretrained_ckpt is not None:
        segan.G.load_pretrained(opts.g_pretrained_ckpt, True)
    if opts.d_pretrained_ckpt is not None:
        segan.D.load_pretrained(opts.d_pretrained_ckpt, True)
    if opts.h5:
        raise NotImplementedError('--h5 datasets are not implemented')
    collate, workers, pin = collate_fn, opts.num_workers, True
    pcm_loader = False
    if opts.synthetic > 0:
        dset = SyntheticSEDataset(opts.synthetic, opts.slice_size, seed=opts.seed)
    elif opts.pcm_shard is not None:
        if opts.preemph_norm or list(opts.random_scale) != [1]:
            raise NotImplementedError('--pcm_shard supports the default pipeline only '
                                      '(no --preemph_norm, no --random_scale)')
        dset = PCMShardDataset(opts.pcm_shard)
        # whole batches gathered by worker processes; the GPU prep kernel runs in this process
        pcm_loader = True
    else:
        dset = SEDataset(opts.clean_trainset, opts.noisy_trainset, opts.preemph,
                         cache_dir=opts.cache_dir, split='train', stride=opts.data_stride,
                         slice_size=opts.slice_size, max_samples=opts.max_samples, verbose=True,
                         preemph_norm=opts.preemph_norm, random_scale=opts.random_scale)
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dset, num_replicas=world, rank=rank, shuffle=True,
                                     seed=opts.seed, drop_last=True)
    if pcm_loader:
        from segan_pytorch_amd.datasets import PCMShardLoader
        dloader = PCMShardLoader(dset, opts.batch_size, opts.preemph, device, sampler=sampler,
                                 drop_last=(world > 1), num_workers=max(1, min(2, opts.num_workers)))
    else:
        dloader = DataLoader(dset, batch_size=opts.batch_size, shuffle=(sampler is None),
                             sampler=sampler, num_workers=workers, pin_memory=pin,
                             collate_fn=collate, drop_last=(world > 1))
    va_dloader = None
    if opts.clean_valset is not None:
        # reference train.py:70-91: one pass over a batch of 300 validation slices per epoch.  The
        # objective here is the segmental SNR, computed on the device (segan_ssnr); the reference's
        # COVL / PESQ terms need the external PESQ binary and are outside the accelerated path.
        va_dset = SEDataset(opts.clean_valset, opts.noisy_valset, opts.preemph,
                            cache_dir=opts.cache_dir, split='valid', stride=opts.data_stride,
                            slice_size=opts.slice_size, max_samples=opts.max_samples, verbose=True,
                            preemph_norm=opts.preemph_norm)
        va_dloader = DataLoader(va_dset, batch_size=300, shuffle=False,
                                num_workers=opts.num_workers, pin_memory=True,
                                collate_fn=collate_fn)
    # per-rank host RNG streams for z and the phase shifts
    random.seed(opts.seed + rank)
    torch.manual_seed(opts.seed + rank)
    try:
        segan.train(opts, dloader, losses.MSELoss(), opts.l1_weight, opts.l1_dec_step,
                    opts.l1_dec_epoch, opts.save_freq, va_dloader=va_dloader, device=device)
    finally:
        sdist.destroy_native()      # the library-owned RCCL communicators (SEGAN_COMM=native)


if __name__ == '__main__':
    opts = build_parser().parse_args()
    opts.bias = not opts.no_bias
    os.makedirs(opts.save_path, exist_ok=True)
    if int(os.environ.get('RANK', '0')) == 0:
        with open(os.path.join(opts.save_path, 'train.opts'), 'w') as cfg_f:
            cfg_f.write(json.dumps(vars(opts), indent=2))
        print('Parsed arguments: ', json.dumps(vars(opts), indent=2))
    main(opts)
