"""Repeats the body of tests/test_gpu_model.py::test_default_segan_plus_step_matches_reference
and prints the worst margin of every check over N runs (run-to-run variation comes from the
order of the fp32 atomics in wgrad / stream-K)."""
import os, sys, random
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import max_rel
import test_gpu_model as T
from oracle import segan_oracle as O
from segan_pytorch_amd.datasets import synthetic_pairs

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
POISON = len(sys.argv) > 2 and sys.argv[2] == 'poison'


def poison():
    """Fill the caching allocator's free blocks with garbage so that any read of
    uninitialised memory shows up as a gross error instead of a rare flake."""
    blocks = [torch.full((n,), 123.456, device='cuda') for n in
              (1 << 28, 1 << 27, 1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 12) * 2]
    del blocks
fx = torch.load(os.path.join(ROOT, 'tests', 'golden', 'segan_plus_b2.pt'))
worst = {}


def rec(name, v):
    worst[name] = max(worst.get(name, 0.0), float(v))


def chk(name, t, c):
    t = t.detach().double().cpu().reshape(-1)
    scale = max(c['abs'], 1e-30)
    rec(name + ':sum', abs(t.sum().item() - c['sum']) / scale)
    rec(name + ':abs', abs(t.abs().sum().item() - c['abs']) / scale)
    got = t[c['sample_idx']].float()
    den = max(c['sample'].abs().max().item(), 1e-30)
    rec(name + ':sample', (got - c['sample']).abs().max().item() / den)


for it in range(N):
    if POISON:
        poison()
    m = T.build(fx, seed=fx['seed'])
    clean, noisy = synthetic_pairs(2, 16384, fx['data_seed'])
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(fx['z_seed']))
    g0 = {k: v.detach().cpu().clone() for k, v in m.G.state_dict().items()}
    with torch.no_grad():
        m.G.train()
        y = m.G(noisy.to('cuda'), z=z.to('cuda'))
    rec('G_out_maxabs', (y.cpu() - fx['Genh']).abs().max().item())
    if POISON:
        poison()
    (d_real_loss, d_fake_loss, g_adv, g_l1), Gopt, Dopt = T.run_step(m, fx, clean, noisy, z)
    for got, key in ((d_real_loss, 'd_real_loss'), (d_fake_loss, 'd_fake_loss'),
                     (g_adv, 'g_adv_loss'), (g_l1, 'g_l1_loss')):
        rec('loss:' + key, max_rel(got, fx[key]))
    dn, gn = dict(m.D.named_parameters()), dict(m.G.named_parameters())
    for k, c in fx['d_grads'].items():
        if not k.endswith('conv.bias'):
            chk('dgrad', dn[k].grad, c)
    for k, v in fx['small_d_grads'].items():
        if not k.endswith('conv.bias'):
            rec('small_d', max_rel(dn[k].grad, v))
    for k, c in fx['g_grads'].items():
        chk('tier2', gn[k].grad, c)
    d_after = {k: v.detach().cpu().clone() for k, v in m.D.state_dict().items()}
    G = {k: v.clone().requires_grad_(True) for k, v in g0.items()}
    st = fx['opts']['genc_poolings']
    genh = O.generator_forward(G, noisy, z, st)
    d = O.discriminator_forward(d_after, torch.cat((genh, noisy), 1), fx['rolls'][2], st)
    loss = F.mse_loss(d.view(-1), torch.ones(2)) + 100.0 * F.l1_loss(genh, clean)
    keys = list(G.keys())
    ograds = torch.autograd.grad(loss, [G[k] for k in keys])
    # the same oracle evaluation a second time: is the CPU side itself reproducible?
    G2 = {k: v.clone().requires_grad_(True) for k, v in g0.items()}
    genh2 = O.generator_forward(G2, noisy, z, st)
    d2 = O.discriminator_forward(d_after, torch.cat((genh2, noisy), 1), fx['rolls'][2], st)
    loss2 = F.mse_loss(d2.view(-1), torch.ones(2)) + 100.0 * F.l1_loss(genh2, clean)
    ograds2 = torch.autograd.grad(loss2, [G2[k] for k in keys])
    orep = max(max_rel(a, b) for a, b in zip(ograds, ograds2))
    print('iter', it, 'g_adv_vs_golden %.2e' % max_rel(g_adv, fx['g_adv_loss']),
          'oracle_loss_vs_gpu %.2e' % max_rel(loss, g_adv + g_l1),
          'oracle_repeat %.2e' % orep, 'genh_oracle_vs_golden %.2e' % (genh.detach() - fx['Genh']).abs().max().item(),
          flush=True)
    for k, g in zip(keys, ograds):
        e = max_rel(gn[k].grad, g)
        rec('tier3', e)
        if e > 1e-4:
            d = (gn[k].grad.detach().cpu() - g).abs()
            thr = 1e-4 * g.abs().max()
            bad = (d > thr).nonzero()
            if k == 'enc_blocks.0.conv.bias':
                print('  BAD', it, k, 'err %.2e' % e, 'elements off:', bad.shape[0], 'of', d.numel(), flush=True)
    print(it, {k: '%.2e' % v for k, v in worst.items()}, flush=True)
