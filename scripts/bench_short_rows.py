"""Short-row conv data gradient (segan_conv1d_dgrad_short, GEMM + col2im) against the T-form
kernel on the deep layers of the SEGAN+ nets at batch 300."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segan_pytorch_amd import ops

dev = torch.device('cuda', 0)
_w = torch.randn(4096, 4096, device=dev)
for _ in range(40):
    _w = torch.tanh(_w @ _w * 1e-4)
torch.cuda.synchronize()


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B = 300
for name, N, M, L in (('enc4', 512, 1024, 64), ('enc3', 256, 512, 256), ('enc4 B=100', 512, 1024, 64)):
    b = 100 if 'B=100' in name else B
    w = torch.randn(M, N, 31, device=dev) * 0.05
    da = torch.randn(b, M, L // 4, device=dev)
    pk = ops.WeightPack()
    gf = 2.0 * b * M * N * 31 * (L // 4) / 1e9
    saved = ops._SHORT_LS
    ops._SHORT_LS = ()
    t_t = timeit(lambda: ops.conv1d_dgrad(da, w, L, 4, pack=pk))
    ops._SHORT_LS = saved
    t_s = timeit(lambda: ops.conv1d_dgrad_short(da, w, L, 4, pack=pk))
    print('{:12s} T-form {:.3f} ms ({:.1f} TF/s)   short {:.3f} ms ({:.1f} TF/s)'.format(
        name, t_t, gf / t_t, t_s, gf / t_s))
