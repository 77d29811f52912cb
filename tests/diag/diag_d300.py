"""D forward+backward at B=300: error of the GPU and of the fp32 CPU oracle against an fp64
CPU evaluation of the same oracle, per parameter gradient (conditioning of the BatchNorm chain)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch, torch.nn.functional as F
import segan_oracle as O
from segan_pytorch_amd.models import Discriminator
from segan_pytorch_amd import losses
B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(5)
D = Discriminator(2, [64, 128, 256, 512, 1024], 31, poolings=[4] * 5, pool_type='none', pool_slen=16, norm_type='bnorm', phase_shift=5)
for n_, p in D.named_parameters():
    if n_.endswith('act.weight'): p.data.uniform_(0.05, 0.3)
    elif n_.endswith('conv.weight'): p.data.normal_(0.0, 0.02)
sd0 = {k: v.detach().clone() for k, v in D.state_dict().items()}
D = D.cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 2, 16384, generator=g) * 2 - 1
rolls = [2, -5, 1, -1, 4]
D.draw_rolls = lambda: list(rolls)
y, _ = D(x[:, :1].contiguous().cuda(), x[:, 1:].contiguous().cuda())
loss = losses.MSELoss()(y.view(-1), 1.0); loss.backward(); torch.cuda.synchronize()
def run(dt):
    sd = {k: (v.clone().to(dt).requires_grad_(True) if torch.is_floating_point(v) and k.split('.')[-1] not in O._BUFFERS else (v.clone().to(dt) if torch.is_floating_point(v) else v.clone())) for k, v in sd0.items()}
    yo = O.discriminator_forward(sd, x.to(dt), rolls, [4] * 5)
    lo = F.mse_loss(yo.view(-1), torch.ones(B, dtype=dt))
    keys = [k for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad]
    return yo, dict(zip(keys, torch.autograd.grad(lo, [sd[k] for k in keys])))
y64, g64 = run(torch.float64)
y32, g32 = run(torch.float32)
def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
def mr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()
print('logit gpu/fp64 %.2e  cpu32/fp64 %.2e' % (mr(y, y64), mr(y32, y64)))
dn = dict(D.named_parameters())
for k in g64:
    print('%-28s gpu %.2e  cpu32 %.2e   |g|max %.2e   L2: gpu/64 %.2e cpu32/64 %.2e gpu/cpu32 %.2e' % (k, mr(dn[k].grad, g64[k]), mr(g32[k], g64[k]), g64[k].abs().max().item(), l2(dn[k].grad, g64[k]), l2(g32[k], g64[k]), l2(dn[k].grad, g32[k])))
