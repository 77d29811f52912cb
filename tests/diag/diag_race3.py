"""Run-to-run determinism of individual backward kernels at the B=2 shapes of the default D."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segan_pytorch_amd import ops
torch.manual_seed(0)
dev = 'cuda'
B = 2
tests = {}
da0 = torch.randn(B, 64, 4096, device=dev); w0 = torch.randn(64, 2, 31, device=dev) * 0.05
tests['enc0_dgrad_tsmall'] = lambda: ops.conv1d_dgrad(da0, w0, 16384, 4, roll=3)
da1 = torch.randn(B, 128, 1024, device=dev); w1 = torch.randn(128, 64, 31, device=dev) * 0.05
tests['enc1_dgrad_mfma'] = lambda: ops.conv1d_dgrad(da1, w1, 4096, 4, roll=-2)
da2 = torch.randn(B, 256, 256, device=dev); w2 = torch.randn(256, 128, 31, device=dev) * 0.05
tests['enc2_dgrad_mfma'] = lambda: ops.conv1d_dgrad(da2, w2, 1024, 4, roll=1)
da4 = torch.randn(B, 1024, 16, device=dev); w4 = torch.randn(1024, 512, 31, device=dev) * 0.05
tests['enc4_dgrad_mfma'] = lambda: ops.conv1d_dgrad(da4, w4, 64, 4, roll=1)
x1 = torch.randn(B, 64, 4096, device=dev)
tests['enc1_fwd'] = lambda: ops.conv1d_fwd(ops.Src(x1), w1, None, 4, roll=2)
xd = torch.randn(B, 128, 4096, device=dev); wd = torch.randn(128, 1, 31, device=dev) * 0.05
tests['dec4_fwd_tsmall'] = lambda: ops.deconv1d_fwd(ops.Src(xd), wd, torch.zeros(1, device=dev), 4, ops.ACT_TANH)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, fn in tests.items():
    ref = fn().clone()
    nbad, worst = 0, 0.0
    for it in range(N):
        junk = torch.randn(1 << 20, device=dev)          # perturb allocator / timing a little
        y = fn()
        d = (y - ref).abs().max().item()
        if d > 0:
            nbad += 1
            worst = max(worst, d)
        del junk
    print('%-22s runs %d  nonidentical %d  worst abs diff %.3e  (max|ref| %.3e)' % (name, N, nbad, worst, ref.abs().max().item()), flush=True)
