"""Does the REAL training loop keep up with the benchmark?  

Runs `train.py`'s own main() — SEGAN.train on the default net, batch 300, RMSprop, z drawn on the
host, batches from an int16 shard through `--pcm_shard` (worker-process gathers, pinned prefetch,
GPU normalise + pre-emphasis) — for a few epochs on a synthetic shard, and compares the steady-state
time per batch with bench.py's step on resident synthetic data.  Checkpoint writes are outside the
per-batch timing (the reference's `btime` excludes them too: model.py:322-348).

    python scripts/train_loop_bench.py [--items 3000] [--epochs 6]  > profiles/rNN_train_loop.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--items', type=int, default=3000)
ap.add_argument('--epochs', type=int, default=6)
ap.add_argument('--batch', type=int, default=300)
ap.add_argument('--tmp', default='/tmp/segan_train_loop')
args = ap.parse_args()

os.makedirs(args.tmp, exist_ok=True)
prefix = os.path.join(args.tmp, 'synthetic')
T = 16384
# a synthetic int16 shard in build_pcm_shard's format (clean row, noisy row, leading sample)
rng = np.random.RandomState(0)
clean = rng.randint(-20000, 20000, size=(args.items, T + 1)).astype(np.int16)
noisy = np.clip(clean.astype(np.int32) + rng.randint(-2000, 2000, size=clean.shape), -32768, 32767).astype(np.int16)
np.stack((clean, noisy), 1).tofile(prefix + '.pcm16')
from segan_pytorch_amd.datasets import SHARD_MAGIC
json.dump({'magic': SHARD_MAGIC, 'n_items': args.items, 'slice_size': T,
           'names': ['utt_{}'.format(i) for i in range(args.items)],
           'slice_idx': [0] * args.items, 'first': [0] * args.items}, open(prefix + '.json', 'w'))

import train
from segan_pytorch_amd.models import core
from segan_pytorch_amd.models import model as M

# steady-state time per batch: the clock starts (after a device synchronisation) at the first
# step of the SECOND epoch and stops, after another synchronisation, when train() returns — the
# host launches ahead of the GPU, so per-step host timers alone would flatter the loop.  No
# checkpoint files (the reference's `btime` excludes their writes too).
per_epoch = args.items // args.batch
orig = M.SEGAN.gan_step
state = {'n': 0, 't0': None}


def counted(self, *a, **k):
    if state['n'] == per_epoch:
        torch.cuda.synchronize()
        state['t0'] = time.perf_counter()
    state['n'] += 1
    return orig(self, *a, **k)


M.SEGAN.gan_step = counted
core.Model.save = lambda self, *a, **k: None
opts = train.build_parser().parse_args(
    ['--pcm_shard', prefix, '--batch_size', str(args.batch), '--epoch', str(args.epochs),
     '--save_path', os.path.join(args.tmp, 'ckpt'), '--no_train_gen', '--save_freq', '1000',
     '--num_workers', '2'])
opts.bias = not opts.no_bias
os.makedirs(opts.save_path, exist_ok=True)
train.main(opts)
torch.cuda.synchronize()
t1 = time.perf_counter()
timed = state['n'] - per_epoch
ms = 1e3 * (t1 - state['t0']) / timed
out = {'what': 'train.py --pcm_shard (default SEGAN+ net, batch {}, RMSprop, host z, int16 shard '
               'through worker gathers + GPU prep): {} epochs of {} batches, the first epoch is '
               'warm-up, device-synchronised clock around the rest'.format(args.batch, args.epochs, per_epoch),
       'ms_per_batch': ms, 'chunks_per_s': args.batch * 1e3 / ms, 'batches_timed': timed}
try:
    import glob
    b = json.load(open(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_bench_line.json')))[-1]))
    out['committed_bench_ms_per_step'] = b['ms_per_step']
except Exception:
    pass
print(json.dumps(out))
