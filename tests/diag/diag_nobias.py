"""Diagnostic: generator-phase gradients of the no-bias default net at B=2, GPU vs CPU oracle
(through the oracle's post-step D), in three arithmetic variants.  If the deviations are
ReLU-gate flips they move around between variants; a kernel bug would not."""
import os, sys, random
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
from types import SimpleNamespace
import segan_oracle as O
from segan_pytorch_amd.models import SEGAN
from segan_pytorch_amd.datasets import synthetic_pairs
from segan_pytorch_amd import ops, losses
name = sys.argv[1] if len(sys.argv) > 1 else 'segan_plus_nobias_b2.pt'
fx = torch.load(os.path.join(ROOT, 'tests', 'golden', name), weights_only=False)
clean, noisy = synthetic_pairs(2, 16384, fx['data_seed'])
clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(fx['z_seed']))
st = fx['opts']['genc_poolings']
ref = None
for prec, det in (('fp32', True), ('fp32', False), ('bf16x3', True)):
    ops.set_precision(prec); ops.set_deterministic(det)
    random.seed(fx['seed']); np.random.seed(fx['seed']); torch.manual_seed(fx['seed'])
    m = SEGAN(SimpleNamespace(**fx['opts']))
    g0 = {k: v.clone() for k, v in m.G.state_dict().items()}
    d0 = {k: v.clone() for k, v in m.D.state_dict().items()}
    if ref is None:
        ref = O.gan_step(g0, d0, clean, noisy, z, fx['rolls'], st, 100.0, 5e-5)
    m = m.to('cuda')
    it = iter(fx['rolls']); m.D.draw_rolls = lambda: list(next(it))
    Gopt, Dopt = m.build_optimizers(SimpleNamespace(**fx['opts']))
    m.G.train(); m.D.train()
    crit = losses.MSELoss()
    cg, ng, zg = clean.cuda(), noisy.cuda(), z.cuda()
    Genh, dr, df = m.d_phase(cg, ng, Dopt, crit, z=zg)
    m.D.load_state_dict({k: ref['D'][k] if k in ref['D'] else v for k, v in d0.items()})
    ops.bump_weights_epoch()
    ga, gl = m.g_phase(Genh, cg, ng, Gopt, crit, 100.0)
    gn = dict(m.G.named_parameters())
    out = []
    for k, g in ref['g_grads'].items():
        a = gn[k].grad.detach().cpu().double(); b = g.double()
        out.append((float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm()), k))
    out.sort(reverse=True)
    print(prec, 'det' if det else 'atomics', 'worst 4 (max_rel, rel_l2, tensor):', [(round(a, 6), round(b, 6), k) for a, b, k in out[:4]])
