"""Accumulation error of the fp32 contractions against fp64, next to torch's own CPU fp32 result
of the same op: relative L2 and max error of forward / data gradient / weight gradient of the
deep layers (the longest contractions).  A/B two builds with SEGAN_HIP_LIB=<path to .so>."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from segan_pytorch_amd import ops
DEV = 'cuda'
torch.backends.mkldnn.enabled = False     # the oracle's setting (oneDNN's transposed conv is off by 0.2 on the GPU box's host)
if len(sys.argv) > 2:
    ops.set_accumulation(sys.argv[2])       # 'plain' | 'blocked'
def rnd(*s, seed=0, scale=1.0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale).float()
def err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double()
    return ((a - b).norm() / b.norm()).item(), ((a - b).abs().max() / b.abs().max()).item()
def show(tag, gpu, c32, ref):
    g, c = err(gpu, ref), err(c32, ref)
    print('%-22s gpu L2 %.2e max %.2e | cpu32 L2 %.2e max %.2e | ratio %.2f' % (tag, g[0], g[1], c[0], c[1], g[0] / c[0]))
S, K = 4, 31
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for name, N, M, L in (('enc3', 256, 512, 256), ('enc4', 512, 1024, 64)):
    x, w, b = rnd(B, N, L, seed=1), rnd(M, N, K, seed=2, scale=0.05), rnd(M, seed=3)
    P = (K // 2 - 1, K // 2)
    outs = {}
    for dt in (torch.float64, torch.float32):
        xd, wd = x.to(dt).requires_grad_(True), w.to(dt).requires_grad_(True)
        y = F.conv1d(F.pad(xd, P, mode='reflect'), wd, b.to(dt), stride=S)
        da = rnd(*y.shape, seed=4)
        y.backward(da.to(dt))
        outs[dt] = (y.detach(), xd.grad, wd.grad)
    xg, wg, dag = x.to(DEV), w.to(DEV), da.to(DEV)
    yg = ops.conv1d_fwd(ops.Src(xg), wg, b.to(DEV), S)
    dx = ops.conv1d_dgrad(dag, wg, L, S)
    dw = torch.zeros_like(wg)
    ops.wgrad(ops.Src(dag), ops.Src(xg), dw, K, S, ops.conv_pad(K, S)[0], ops.PAD_REFLECT)
    for t, g_, i in (('fwd', yg, 0), ('dgrad', dx, 1), ('wgrad', dw, 2)):
        show('%s %s K=%d' % (name, t, (N * K, M * 8, B * L // 4)[i]), g_, outs[torch.float32][i], outs[torch.float64][i])
for name, M, N, Ls in (('dec0', 2048, 512, 16), ('dec1', 1024, 256, 64)):
    x, w, b = rnd(B, M, Ls, seed=5), rnd(M, N, K, seed=6, scale=0.05), rnd(N, seed=7)
    pad = ops.deconv_pad(K, S)
    outs = {}
    for dt in (torch.float64, torch.float32):
        xd, wd = x.to(dt).requires_grad_(True), w.to(dt).requires_grad_(True)
        y = F.conv_transpose1d(xd, wd, b.to(dt), stride=S, padding=pad)[:, :, :S * Ls]
        dy = rnd(*y.shape, seed=8)
        y.backward(dy.to(dt))
        outs[dt] = (y.detach(), xd.grad, wd.grad)
    xg, wg, dyg = x.to(DEV), w.to(DEV), dy.to(DEV)
    yg = ops.deconv1d_fwd(ops.Src(xg), wg, b.to(DEV), S)
    dx0, dx1 = ops.deconv1d_dgrad(dyg, wg, S, 0)
    dw = torch.zeros_like(wg)
    ops.wgrad(ops.Src(xg), ops.Src(dyg), dw, K, S, pad, ops.PAD_ZERO)
    for t, g_, i in (('fwd', yg, 0), ('dgrad', dx1, 1), ('wgrad', dw, 2)):
        show('%s %s K=%d' % (name, t, (M * 8, N * K, B * Ls)[i]), g_, outs[torch.float32][i], outs[torch.float64][i])
