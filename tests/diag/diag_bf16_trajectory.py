"""Diagnostic behind tests/test_gpu_bf16_parity.py::test_bf16_trajectory_tracks_fp32_over_20_steps:
the loss curves of 20 GAN steps from identical state in fp32, bf16x3 (fp32-class rounding, a
different summation: the CONTROL for how far two correct fp32-class runs drift apart) and bf16,
at two batch sizes.  python tests/diag/diag_bf16_trajectory.py > out.json"""
import json
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from segan_pytorch_amd import losses, ops  # noqa: E402
from segan_pytorch_amd.datasets import synthetic_pairs  # noqa: E402
from segan_pytorch_amd.models import SEGAN  # noqa: E402

DEV = 'cuda'
STEPS = 20
out = {}
for B in (8, 32):
    opts = bench.default_opts()
    random.seed(111); np.random.seed(111); torch.manual_seed(111)
    m0 = SEGAN(SimpleNamespace(**opts))
    gsd0 = {k: v.detach().clone() for k, v in m0.G.state_dict().items()}
    dsd0 = {k: v.detach().clone() for k, v in m0.D.state_dict().items()}
    del m0
    clean, noisy = synthetic_pairs(B, 16384, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    zs = [torch.randn(B, 1024, 16, generator=torch.Generator().manual_seed(100 + i)) for i in range(STEPS)]
    rng = random.Random(5)
    rolls = [[[rng.randint(1, 5) * (1 if rng.random() > 0.5 else -1) for _ in range(5)] for _ in range(3)]
             for _ in range(STEPS)]
    res = {}
    for prec in ('fp32', 'bf16x3', 'bf16'):
        ops.set_precision(prec)
        m = SEGAN(SimpleNamespace(**opts))
        m.G.load_state_dict(gsd0); m.D.load_state_dict(dsd0)
        m = m.to(DEV)
        Gopt, Dopt = m.build_optimizers(SimpleNamespace(**opts))
        m.G.train(); m.D.train()
        flat = iter([r for step in rolls for r in step])
        m.D.draw_rolls = lambda: list(next(flat))
        cg, ng = clean.to(DEV), noisy.to(DEV)
        log = []
        for i in range(STEPS):
            o = m.gan_step(cg, ng, Gopt, Dopt, losses.MSELoss(), 100.0, z=zs[i].to(DEV))
            log.append([float(v) for v in o])
        g = torch.cat([(v.detach().cpu().double() - gsd0[k].double()).flatten() for k, v in m.G.state_dict().items()])
        d = torch.cat([(v.detach().cpu().double() - dsd0[k].double()).flatten() for k, v in m.D.state_dict().items()
                       if torch.is_floating_point(v) and k.split('.')[-1] not in ('running_mean', 'running_var')])
        res[prec] = dict(log=log, g=g, d=d)
        ops.set_precision('fp32')
        del m, Gopt, Dopt
    o = {'curves': {p: res[p]['log'] for p in res}}
    for p in ('bf16x3', 'bf16'):
        for n in ('g', 'd'):
            a, b = res['fp32'][n], res[p][n]
            o['{}_{}_cos'.format(p, n)] = float(torch.dot(a, b) / (a.norm() * b.norm()))
            o['{}_{}_norm_ratio'.format(p, n)] = float(b.norm() / a.norm())
    out['B{}'.format(B)] = o
print(json.dumps(out))
