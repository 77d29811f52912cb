"""Is the oracle a fair stand-in for the reference as the CPU baseline?  TEST INFRASTRUCTURE ONLY.

`bench.py`'s `cpu_baseline` times `oracle/segan_oracle.gan_step` (kind = "port") because the GPU
box has no /root/reference.  This script — runnable only in the build container — times the
REAL reference's literal `SEGAN.train` (segan/models/model.py:271-348, which prints its own
per-batch `btime`, model.py:322-348) and the oracle step back to back on the same host, same
weights, same batches, same thread count, for both oneDNN settings, and writes the ratio:

    python oracle/time_ref_vs_port.py [B=32] [steps=3] > profiles/rNN_ref_vs_port_cpu.json

The reference's step includes the discriminator weight gradients of the generator phase, which
its next `Dopt.zero_grad()` discards (model.py:315-320); the oracle's autograd.grad call never
computes them, so the oracle is expected to be slightly FASTER than the reference, i.e. the
"port" baseline is the harder denominator.
"""
import contextlib
import io
import json
import os
import random
import re
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
import segan_oracle as O  # noqa: E402
from make_golden import base_opts, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ref = ref_harness.import_reference()
    o = base_opts()
    o.update(dict(batch_size=B, epoch=1, save_path='/tmp/segan_time_ckpt', save_freq=10 ** 9,
                  no_train_gen=True))
    rows = {}
    for onednn in (False, True):
        random.seed(111); np.random.seed(111); torch.manual_seed(111)
        segan = ref.SEGAN(SimpleNamespace(**o))
        gsd0 = {k: v.detach().clone() for k, v in segan.G.state_dict().items()}
        dsd0 = {k: v.detach().clone() for k, v in segan.D.state_dict().items()}
        batches = [synth(B, 16384, s) for s in range(steps + 1)]          # first = warm-up
        loader = [[['u'] * B, c, n, torch.zeros(B)] for c, n in batches]
        torch.backends.mkldnn.enabled = onednn
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            segan.train(SimpleNamespace(**o), loader, nn.MSELoss(), o['l1_weight'], o['l1_dec_step'],
                        o['l1_dec_epoch'], 1, va_dloader=None, device='cpu')
        wall_ref = time.perf_counter() - t0
        btimes = [float(x) for x in re.findall(r'(?<!m)btime: ([0-9.]+) s', buf.getvalue())]
        # the oracle on the same weights / batches; z drawn like the reference draws it (CPU randn)
        st = o['genc_poolings']
        rolls = [[1, -2, 3, -4, 5], [-3, 2, -1, 5, 4], [2, -5, 1, -1, -4]]
        gsd, dsd, g_sq, d_sq = gsd0, dsd0, None, None
        ptimes = []
        for c, n in batches:
            t0 = time.perf_counter()
            z = torch.randn(B, 1024, 16)
            res = O.gan_step(gsd, dsd, c.unsqueeze(1), n.unsqueeze(1), z, rolls, st, 100.0, 5e-5,
                             g_sq=g_sq, d_sq=d_sq)
            ptimes.append(time.perf_counter() - t0)
            gsd, dsd, g_sq, d_sq = res['G'], res['D'], res['g_sq'], res['d_sq']
        rows['onednn_on' if onednn else 'onednn_off'] = {
            'reference_btime_s': btimes, 'port_step_s': ptimes,
            'reference_mean_steps_ge2': float(np.mean(btimes[1:])),
            'port_mean_steps_ge2': float(np.mean(ptimes[1:])),
            'port_over_reference': float(np.mean(ptimes[1:]) / np.mean(btimes[1:])),
            'reference_train_wall_s': wall_ref}
    torch.backends.mkldnn.enabled = False
    print(json.dumps({
        'what': 'the reference\'s literal SEGAN.train (its own btime print, model.py:322-348) and the '
                'oracle\'s gan_step (bench.py cpu_baseline kind="port") back to back on the build '
                'container\'s host; step 1 of each is a warm-up and excluded from the means',
        'batch': B, 'timed_steps': steps, 'threads': torch.get_num_threads(), 'nproc': os.cpu_count(),
        'torch': torch.__version__, 'rows': rows}, indent=1))


if __name__ == '__main__':
    main()
