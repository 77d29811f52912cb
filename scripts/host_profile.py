"""Where the HOST time of a GAN step goes (round-5 review, weak 4: the 11-layer shape and the WSEGAN
step were host-bound on some boxes).  For one workload:

* `enqueue_ms`: wall time of one_step() on the host when the GPU starts idle (device synchronised
  before, NOT after): with no hidden sync inside the step this is the pure launch cost — python,
  ctypes, the caching allocator, hipLaunchKernel — and the GPU is never the one being waited for
  unless the launch queue fills;
* `step_ms`: the same steps back to back, synchronised at the end only (what bench.py times);
* `gpu_ms`: sum of the kernels' durations is not available without a tracer — the step time with the
  host far ahead stands in for it; `enqueue_ms / step_ms` near or above 1 means host-bound;
* `--cprofile N`: cProfile over N steps, top functions by cumulative and by own time.

    python scripts/host_profile.py --shape vanilla11
    python scripts/host_profile.py --wsegan
"""
import argparse
import cProfile
import io
import json
import os
import pstats
import random
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from segan_pytorch_amd import losses, ops
from segan_pytorch_amd.datasets import synthetic_pairs
from segan_pytorch_amd.models import SEGAN, WSEGAN

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='segan_plus', choices=['segan_plus', 'vanilla11'])
ap.add_argument('--wsegan', action='store_true')
ap.add_argument('--precision', default='fp32')
ap.add_argument('--batch', type=int, default=300)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--cprofile', type=int, default=3)
ap.add_argument('--device-z', action='store_true')
ap.add_argument('--no-prefetch', action='store_true')
args = ap.parse_args()

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
ops.set_precision(args.precision)
opts = bench.default_opts()
if args.shape == 'vanilla11':
    opts.update(bench.VANILLA11)
random.seed(111); np.random.seed(111); torch.manual_seed(111)
if args.wsegan:
    opts.update(dict(misalign_pair=True, interf_pair=False, pow_weight=0.001, vanilla_gan=False, n_fft=2048))
    model = WSEGAN(SimpleNamespace(**opts)).to(dev)
else:
    model = SEGAN(SimpleNamespace(**opts)).to(dev)
Gopt, Dopt = model.build_optimizers(SimpleNamespace(**opts))
model.G.train(); model.D.train()
crit = losses.MSELoss()
B = args.batch
clean, noisy = synthetic_pairs(B, 16384, seed=0, device=dev)
clean, noisy = clean.unsqueeze(1).contiguous(), noisy.unsqueeze(1).contiguous()
names = ['utt_additive_{}'.format(i) if i % 2 == 0 else 'utt_{}'.format(i) for i in range(B)]
if args.device_z:
    model.G.z_generator = torch.Generator(device=dev).manual_seed(0)
else:
    model.G.z_prefetch = not args.no_prefetch and bench.z_lookahead_ok(args.wsegan)


def one_step():
    if args.wsegan:
        return model.wgan_step(names, clean, noisy, Gopt, Dopt, 100.0, z=None)
    return model.gan_step(clean, noisy, Gopt, Dopt, crit, 100.0, z=None)


for _ in range(3):
    one_step()
torch.cuda.synchronize()
enq = []
for _ in range(args.steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one_step()
    enq.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    one_step()
torch.cuda.synchronize()
step_ms = 1e3 * (time.perf_counter() - t0) / args.steps

# launches per step through the C ABI (python-side count of library calls)
lib = ops._lib.load()
calls = {}
res = {'workload': ('wsegan ' if args.wsegan else '') + args.shape, 'precision': args.precision,
       'batch': B, 'enqueue_ms_per_step': {'median': sorted(enq)[len(enq) // 2], 'min': min(enq),
                                           'max': max(enq), 'all': [round(e, 2) for e in enq]},
       'step_ms': step_ms, 'host_share_of_step': sorted(enq)[len(enq) // 2] / step_ms,
       'cpus': len(os.sched_getaffinity(0)), 'torch_threads': torch.get_num_threads()}
print(json.dumps(res))
if args.cprofile > 0:
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(args.cprofile):
        one_step()
    pr.disable()
    torch.cuda.synchronize()
    for key in ('cumulative', 'tottime'):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print('---- cProfile over {} steps, by {} ----'.format(args.cprofile, key))
        print(s.getvalue()[:9000])
