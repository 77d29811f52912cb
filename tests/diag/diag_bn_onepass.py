import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from segan_pytorch_amd import ops
def rnd(*s, seed=0, scale=1.0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
for (B,C,L,mu,sd) in [(300, 8, 4096, 10.0, 0.1), (300, 64, 16, -3.0, 0.02),(7, 5, 1001, 0.5, 2.0), (80, 16, 256, 50.0, 1.0)]:
    x = (rnd(B, C, L, seed=31) * sd + mu + rnd(C, seed=32).view(1, C, 1) * sd).float()
    xd=x.double(); mr=xd.mean((0,2)); vr=xd.var((0,2),unbiased=False)
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    mean, rstd, scale, shift = ops.bn_stats(x.cuda(), None, None, 0.0, 1.0, rm, rv)
    var = 1.0/(rstd.double().cpu()**2)
    # fp32 two-pass on CPU for comparison
    m32 = x.mean((0,2)); v32 = ((x-m32.view(1,-1,1))**2).mean((0,2))
    print((B,C,L,mu,sd), 'mean err/std', float(((mean.double().cpu()-mr).abs()/vr.sqrt()).max()), 'var rel', float(((var-vr).abs()/vr).max()), 'cpu fp32 two-pass var rel', float(((v32.double()-vr).abs()/vr).max()))
