"""Measured error of the three contraction precisions against an fp64 CPU computation on a
mid-layer shape (conv 128->256, k31, s4, L=1024, B=4): max |err| / max |ref| and RMS error."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segan_pytorch_amd import ops

torch.manual_seed(0)
B, N, M, L, K, S = 4, 128, 256, 1024, 31, 4
x = torch.randn(B, N, L)
w = torch.randn(M, N, K) * 0.02
xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
ref = F.conv1d(F.pad(xd, (K // 2 - 1, K // 2), mode='reflect'), wd, None, stride=S)
da = torch.randn(*ref.shape)
ref.backward(da.double())


def err(got, want):
    d = got.double().cpu() - want
    return d.abs().max().item() / want.abs().max().item(), (d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()


print('%-8s %-6s %12s %12s' % ('mode', 'op', 'max-rel', 'rms-rel'))
for mode in ('fp32', 'bf16x3', 'bf16'):
    ops.set_precision(mode)
    out = ops.conv1d_fwd(ops.Src(x.cuda()), w.cuda(), None, S)
    dx = ops.conv1d_dgrad(da.cuda(), w.cuda(), L, S)
    dw = torch.zeros(M, N, K, device='cuda')
    ops.wgrad(ops.Src(da.cuda()), ops.Src(x.cuda()), dw, K, S, K // 2 - 1, ops.PAD_REFLECT)
    for name, got, want in (('fwd', out, ref.detach()), ('dgrad', dx, xd.grad), ('wgrad', dw, wd.grad)):
        a, b = err(got, want)
        print('%-8s %-6s %12.3e %12.3e' % (mode, name, a, b))
ops.set_precision('fp32')
