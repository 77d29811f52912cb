"""Diagnostic (not a test): for the two default-net fixtures at B=2, count the ReLU gates (PReLU
slope 0 at init) on which the GPU forward and the CPU oracle disagree — in G, and in D on the
fake pair — and how far the pre-activations at those gates are from zero."""
import os, sys, random
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
from types import SimpleNamespace
import segan_oracle as O
from segan_pytorch_amd.models import SEGAN
from segan_pytorch_amd.datasets import synthetic_pairs
from segan_pytorch_amd import ops
ops.set_deterministic(True)
for name in ('segan_plus_b2.pt', 'segan_plus_nobias_b2.pt'):
    fx = torch.load(os.path.join(ROOT, 'tests', 'golden', name), weights_only=False)
    random.seed(fx['seed']); np.random.seed(fx['seed']); torch.manual_seed(fx['seed'])
    m = SEGAN(SimpleNamespace(**fx['opts']))
    g0 = {k: v.clone() for k, v in m.G.state_dict().items()}
    d0 = {k: v.clone() for k, v in m.D.state_dict().items()}
    m = m.to('cuda')
    clean, noisy = synthetic_pairs(2, 16384, fx['data_seed'])
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(fx['z_seed']))
    st = fx['opts']['genc_poolings']
    m.G.train(); m.D.train()
    with torch.no_grad():
        y, hall = m.G(noisy.cuda(), z=z.cuda(), ret_hid=True)
        yc, hc = O.generator_forward(g0, noisy, z, st, ret_hid=True)
        for k in hc:
            if k.startswith('enc_') and k != 'enc_zc' or (k.startswith('dec_') and k != 'dec_4'):
                a, b = hall[k].cpu(), hc[k]
                flips = ((a > 0) != (b > 0))
                print(name, 'G', k, 'flips', int(flips.sum()), 'of', a.numel(),
                      'max |h| at flips', float(torch.maximum(a.abs(), b.abs())[flips].max()) if flips.any() else 0.0)
        rolls = fx['rolls'][2]
        m.D.draw_rolls = lambda: list(rolls)
        yd, acts = m.D(y, noisy.cuda())
        dd = {k: v.clone() for k, v in d0.items()}
        ydc, ac = O.discriminator_forward(dd, torch.cat((yc, noisy), 1), rolls, st, ret_act=True)
        for k in ac:
            if k.startswith('h_'):
                a, b = acts[k].cpu(), ac[k]
                flips = ((a > 0) != (b > 0))
                print(name, 'D', k, 'flips', int(flips.sum()), 'of', a.numel(),
                      'max |h| at flips', float(torch.maximum(a.abs(), b.abs())[flips].max()) if flips.any() else 0.0,
                      'max_abs diff', float((a - b).abs().max()))
