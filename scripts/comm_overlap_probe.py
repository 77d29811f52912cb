"""What a resident collective costs the step, measured on ONE GPU (round-4 review,
weak point 7: "how RCCL's workgroups get CUs beside the persistent contraction kernels is an
argument, not a measurement").

The data-parallel reducer runs in a one-rank group (SEGAN_DP_SINGLE=1: buckets, arming, the
gradient-ready reports from inside the backward passes — the production code), but instead of the
one-rank all-reduce (a no-op) every bucket launches scripts/comm_slot_probe.hip on a HIGH-PRIORITY
side stream behind the kernels that produced it: `nchan` workgroups of 256 threads that hold their
slots for bytes / bandwidth — the footprint and the duration a link-bound all-reduce of that bucket
would have at 8 GPUs.  The compute stream waits for them at the optimizer step, as for real ones.

Timed: the full GAN step at batch 300 (a) without any emulated collective, (b) with them, (c) with
them and the launch planners told to leave `nchan` slots free (ops.set_reserved_slots), (d) the
reserve alone.   usage: python scripts/comm_overlap_probe.py > gpurun_out/comm_overlap.json
"""
import ctypes
import json
import os
import random
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(SEGAN_DP_SINGLE='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0',
                  MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
import numpy as np
import torch

import bench
from segan_pytorch_amd import distributed as sdist
from segan_pytorch_amd import losses, ops
from segan_pytorch_amd.datasets import synthetic_pairs
from segan_pytorch_amd.models import SEGAN

so = os.path.join(ROOT, 'scripts', 'libcommslot.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so,
                       os.path.join(ROOT, 'scripts', 'comm_slot_probe.hip')])
lib = ctypes.CDLL(so)
lib.comm_slot_launch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_ulonglong,
                                 ctypes.c_void_p, ctypes.c_void_p]

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
sdist.init_from_env()
assert sdist._collectives_on()
opts = bench.default_opts()
random.seed(111); np.random.seed(111); torch.manual_seed(111)
model = SEGAN(SimpleNamespace(**opts)).to(dev)
Gopt, Dopt = model.build_optimizers(SimpleNamespace(**opts))
model.G.train(); model.D.train()
B = 300
clean, noisy = synthetic_pairs(B, 16384, seed=0, device=dev)
clean, noisy = clean.unsqueeze(1).contiguous(), noisy.unsqueeze(1).contiguous()
model.G.z_generator = torch.Generator(device=dev).manual_seed(0)     # z off the host: only the overlap is measured
crit = losses.MSELoss()

NCHAN = 32
side = torch.cuda.Stream(priority=-1)
scratch = torch.zeros(NCHAN * 65536, device=dev)                      # 256 KB per workgroup and pass
stamps = torch.zeros(2 * NCHAN * 64, dtype=torch.int64, device=dev)
state = {'gbs': None, 'launched': 0, 'events': []}
orig_send, orig_finish = sdist.GradReducer._send, sdist.GradReducer.finish


def emu_send(self, b):
    lo, hi = self.buckets[b]
    if state['gbs'] is not None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)                                           # behind the kernels that produced the bucket
        ticks = int((hi - lo) * 4 / (state['gbs'] * 1e9) * 1e8)       # 100 MHz ticks the collective would last
        k = state['launched'] % 64
        lib.comm_slot_launch(ctypes.c_void_p(scratch.data_ptr()), 65536, NCHAN, ticks,
                             ctypes.c_void_p(stamps.data_ptr() + 16 * NCHAN * k), ctypes.c_void_p(side.cuda_stream))
        done = torch.cuda.Event()
        done.record(side)
        state['events'].append(done)
        state['launched'] += 1
    self.sent[b] = True


def emu_finish(self):
    for ev in state['events']:
        torch.cuda.current_stream().wait_event(ev)                    # the optimizer step follows the collectives
    state['events'] = []
    return orig_finish(self)


sdist.GradReducer._send = emu_send
sdist.GradReducer.finish = emu_finish


def run(label, gbs, reserve, steps=10, warm=3):
    state['gbs'] = gbs
    ops.set_reserved_slots(reserve)
    for _ in range(warm):
        model.gan_step(clean, noisy, Gopt, Dopt, crit, 100.0)
    torch.cuda.synchronize()
    n0 = state['launched']
    t0 = time.perf_counter()
    for _ in range(steps):
        model.gan_step(clean, noisy, Gopt, Dopt, crit, 100.0)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return {'mode': label, 'emulated_link_gb_per_s': gbs, 'reserved_slots': reserve, 'ms_per_step': ms,
            'collectives_per_step': (state['launched'] - n0) / steps}


rows = []
for rep in range(2):
    rows.append(run('no collectives', None, 0))
    rows.append(run('ring-bound collectives (bucket bytes / 88 GB/s), no reserve', 88.0, 0))
    rows.append(run('ring-bound collectives, planners leave 32 slots free', 88.0, NCHAN))
    rows.append(run('all-link collectives (bucket bytes / 600 GB/s), no reserve', 600.0, 0))
    rows.append(run('all-link collectives, planners leave 32 slots free', 600.0, NCHAN))
    rows.append(run('no collectives, planners leave 32 slots free', None, NCHAN))
ops.set_reserved_slots(0)
r = sdist._reducers
print(json.dumps({
    'what': 'full SEGAN+ GAN step at batch 300, one GPU; every gradient bucket of the data-parallel reducer launches a '
            '32-workgroup x 256-thread kernel on a high-priority side stream that holds its slots for bucket bytes / '
            'the stated bandwidth (scripts/comm_slot_probe.hip); z drawn on the GPU',
    'buckets_mb': {('G' if x.opt is Gopt else 'D'): [round((hi - lo) * 4 / 2 ** 20, 2) for lo, hi in x.buckets]
                   for x in r.values()},
    'rows': rows}, indent=1))
