"""Fresh model + one full GAN step per iteration (fresh weight packs, fresh optimizers), all
seeds fixed: every quantity must repeat to fp32-atomics noise.  Prints which ones deviate."""
import os, sys, random, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import test_gpu_model as T
from segan_pytorch_amd.datasets import synthetic_pairs

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
fx = torch.load(os.path.join(ROOT, 'tests', 'golden', 'segan_plus_b2.pt'))
clean, noisy = synthetic_pairs(2, 16384, fx['data_seed'])
clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(fx['z_seed']))
m0 = T.build(fx, seed=fx['seed'])
sdG = {k: v.clone() for k, v in m0.G.state_dict().items()}
sdD = {k: v.clone() for k, v in m0.D.state_dict().items()}
del m0


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


ref = None
for it in range(N):
    from segan_pytorch_amd.models import SEGAN
    from types import SimpleNamespace
    m = SEGAN(SimpleNamespace(**fx['opts']))
    m.G.load_state_dict(sdG); m.D.load_state_dict(sdD)
    m = m.to('cuda')
    cap = {}
    orig = m.infer_G
    def infer_G(*a, **k):
        y = orig(*a, **k)
        if y.requires_grad:
            y.register_hook(lambda g: cap.__setitem__('dGenh', g.detach().clone()))
        return y
    m.infer_G = infer_G
    out, Gopt, Dopt = T.run_step(m, fx, clean, noisy, z)
    cur = {'dGenh': cap['dGenh']}
    cur.update({'loss%d' % i: o.detach().clone() for i, o in enumerate(out)})
    cur.update({'G.' + k: p.grad.detach().clone() for k, p in m.G.named_parameters()})
    cur.update({'Dw.' + k: p.detach().clone() for k, p in m.D.named_parameters() if k.endswith('conv.weight')})
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        continue
    errs = {k: rel(cur[k], ref[k]) for k in cur}
    bad = {k: '%.1e' % e for k, e in errs.items() if e > 1e-4}
    d = (cur['dGenh'] - ref['dGenh']).flatten()
    big = (d.abs() > 1e-3 * ref['dGenh'].abs().max()).nonzero().flatten()
    print(it, 'max %.2e' % max(errs.values()), 'dGenh %.2e' % errs['dGenh'], 'n_off', big.numel(),
          'idx', big[:6].tolist(), 'delta', [float('%.3e' % v) for v in d[big[:6]].tolist()],
          'ref', [float('%.3e' % v) for v in ref['dGenh'].flatten()[big[:6]].tolist()],
          'absmax %.3e' % ref['dGenh'].abs().max().item(),
          'losses_bad', [k for k in bad if k.startswith('loss')], flush=True)
