import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import torch, torch.nn.functional as F
from segan_pytorch_amd import ops
torch.manual_seed(0)
for (B, N, M, L, roll) in ((80, 64, 128, 4096, 3), (7, 64, 128, 4096, 3), (80, 64, 128, 4096, 0)):
    S, K = 4, 31
    da = torch.randn(B, M, L // S)
    w = torch.randn(M, N, K) * 0.05
    x = torch.zeros(B, N, L, dtype=torch.float64, requires_grad=True)
    xr = torch.roll(x, roll, 2) if roll else x
    y = F.conv1d(F.pad(xr, (14, 15), mode='reflect'), w.double(), None, stride=S)
    y.backward(da.double())
    dx = ops.conv1d_dgrad(da.cuda(), w.cuda(), L, S, roll=roll).cpu().double()
    err = (dx - x.grad).abs()
    bad = (err > 1e-3).nonzero()
    print((B, N, M, L, roll), 'launch', ops.last_corr_launch(), 'bad', len(bad))
    if len(bad):
        print('samples b', sorted(set(bad[:, 0].tolist()))[:20])
        print('channels n', sorted(set(bad[:, 1].tolist()))[:40])
        ii = sorted(set(bad[:, 2].tolist()))
        print('positions i (first 40)', ii[:40], '... last', ii[-10:], 'count', len(ii))
