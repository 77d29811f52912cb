"""Diagnostic: G-phase gradients on the GPU vs the CPU oracle fed with the GPU's own
post-step discriminator weights (separates kernel error from RMSprop sensitivity)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/tests', ROOT + '/oracle'):
    sys.path.insert(0, p)
import random
import torch
import torch.nn.functional as F
from conftest import load_golden
from test_gpu_model import build, run_step
from segan_pytorch_amd.datasets import synthetic_pairs
import segan_oracle as O

fx = load_golden('segan_plus_b2.pt')
m = build(fx, seed=fx['seed'])
g0 = {k: v.detach().cpu().clone() for k, v in m.G.state_dict().items()}
clean, noisy = synthetic_pairs(2, 16384, 0)
clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(0))
out, Gopt, Dopt = run_step(m, fx, clean, noisy, z)
d_after = {k: v.detach().cpu().clone() for k, v in m.D.state_dict().items()}

def rel(t, c):
    t = t.detach().double().cpu().reshape(-1)
    got = t[c['sample_idx']].float()
    return (got - c['sample']).abs().max().item(), c['sample'].abs().max().item()
print('--- D weights after the step vs reference (abs err on 257 samples, max |ref|)')
for k, c in fx['D_after'].items():
    if torch.is_floating_point(d_after[k]):
        e, mx = rel(d_after[k], c)
        print('D_after %-30s abs %.2e (max %.2e)' % (k, e, mx))
# oracle G phase with the GPU's post-step D
G = {k: v.clone().requires_grad_(True) for k, v in g0.items()}
genh = O.generator_forward(G, noisy, z, [4] * 5)
dsd = {k: v.clone() for k, v in d_after.items()}
d = O.discriminator_forward(dsd, torch.cat((genh, noisy), 1), fx['rolls'][2], [4] * 5)
loss = F.mse_loss(d.view(-1), torch.ones(2)) + 100.0 * F.l1_loss(genh, clean)
keys = list(G.keys())
gr = torch.autograd.grad(loss, [G[k] for k in keys])
gn = dict(m.G.named_parameters())
print('--- G grads: GPU vs oracle(with GPU post-step D)')
for k, g in zip(keys, gr):
    a = gn[k].grad.detach().cpu()
    print('G %-30s rel %.2e' % (k, ((a - g).abs().max() / g.abs().max()).item()))
