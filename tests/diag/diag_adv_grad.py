"""Diagnostic: gradient of the adversarial loss w.r.t. the generator output (through D)
on the GPU vs the CPU oracle, using the SAME (GPU-side) discriminator weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/tests', ROOT + '/oracle'):
    sys.path.insert(0, p)
import random
import torch
import torch.nn.functional as F
from conftest import load_golden
from test_gpu_model import build
from segan_pytorch_amd import losses
from segan_pytorch_amd.datasets import synthetic_pairs
import segan_oracle as O

fx = load_golden('segan_plus_b2.pt')
m = build(fx, seed=fx['seed'])
clean, noisy = synthetic_pairs(2, 16384, 0)
clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
g = torch.Generator().manual_seed(0)
genh = torch.tanh(torch.randn(2, 1, 16384, generator=g))
m.D.train()
rolls = [2, -1, 3, -4, 5]
m.D.draw_rolls = lambda: rolls
for frozen in (False, True):
    x = genh.cuda().requires_grad_(True)
    for p in m.D.parameters():
        p.grad = None
    if frozen:
        ps = [p for p in torch.nn.Module.parameters(m.D)]
        for p in ps: p.requires_grad_(False)
    sd = {k: v.detach().cpu().clone() for k, v in m.D.state_dict().items()}
    d, _ = m.D(torch.cat((x, noisy.cuda()), 1))
    loss = losses.MSELoss()(d.view(-1), 1.0)
    loss.backward()
    if frozen:
        for p in ps: p.requires_grad_(True)
    xc = genh.clone().requires_grad_(True)
    dc = O.discriminator_forward(sd, torch.cat((xc, noisy), 1), rolls, [4] * 5)
    lc = F.mse_loss(dc.view(-1), torch.ones(2))
    lc.backward()
    a, b = x.grad.cpu(), xc.grad
    print('frozen', frozen, 'loss', loss.item(), lc.item(), 'adv grad max', b.abs().max().item(),
          'abs err', (a - b).abs().max().item(), 'rel', ((a - b).abs().max() / b.abs().max()).item())
    # where is the error?
    e = (a - b).abs()[0, 0]
    idx = torch.topk(e, 8).indices.sort().values
    print(' worst idx', idx.tolist(), 'err', e[idx].tolist())
    print(' head err', e[:20].tolist())
    print(' tail err', e[-20:].tolist())
