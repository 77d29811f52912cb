"""Determinism hunt: repeat the D dgrad chain (frozen D, as in the generator phase) and the G
backward on fixed inputs and report any run that deviates from the first by more than fp32
atomics noise.  Run it while another process loads the GPU."""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import test_gpu_model as T
from segan_pytorch_amd.datasets import synthetic_pairs
from segan_pytorch_amd.models.model import _frozen

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
fx = torch.load(os.path.join(ROOT, 'tests', 'golden', 'segan_plus_b2.pt'))
m = T.build(fx, seed=fx['seed'])
m.G.train(); m.D.train()
clean, noisy = synthetic_pairs(2, 16384, fx['data_seed'])
clean, noisy = clean.unsqueeze(1).cuda(), noisy.unsqueeze(1).cuda()
z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(fx['z_seed'])).cuda()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


ref_dx = ref_g = None
for it in range(N):
    # ---- D dgrad chain ----
    x = (noisy * 0.7).detach().clone().requires_grad_(True)
    random.seed(5)
    with _frozen(m.D):
        y, _ = m.D(x, noisy)
        (y.view(-1) - 1).pow(2).mean().backward()
    dx = x.grad.detach().clone()
    # ---- G backward ----
    for p in m.G.parameters():
        p.grad = None
    yg = m.G(noisy, z=z)
    (yg * clean).sum().backward()
    g = {k: p.grad.detach().clone() for k, p in m.G.named_parameters()}
    torch.cuda.synchronize()
    if ref_dx is None:
        ref_dx, ref_g = dx, g
        continue
    e_d = rel(dx, ref_dx)
    e_g = max(rel(g[k], ref_g[k]) for k in g)
    flag = 'BAD' if (e_d > 1e-4 or e_g > 1e-4) else ''
    if flag or it % 10 == 0:
        worst = max(g, key=lambda k: rel(g[k], ref_g[k]))
        print(it, 'D_dx %.2e' % e_d, 'G_grads %.2e' % e_g, worst, flag, flush=True)
