"""Per-layer timing of the contraction kernels at the BASELINE shapes (B=300, SEGAN+
default net): conv fwd / dgrad / wgrad for the encoders, deconv fwd / dgrad / wgrad for
the decoder.  HIP events on the launch stream, median of N runs."""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segan_pytorch_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=300)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--only', default='')
ap.add_argument('--shape', default='segan_plus', choices=['segan_plus', 'vanilla11'])
ap.add_argument('--verbose', action='store_true')
args = ap.parse_args()
B, K, S = args.batch, 31, 4
dev = 'cuda'


def timeit(fn, iters=args.iters):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


# the first launches after idle run at a lower clock: warm up before timing anything
_w = torch.randn(4096, 4096, device=dev)
for _ in range(60):
    _w = torch.tanh(_w @ _w * 1e-4)
torch.cuda.synchronize()

rows = []
def report(name, flops, ms):
    rows.append((name, flops / 1e9, ms, flops / ms / 1e9))
    info = ops.last_wgrad_launch() if 'wgrad' in name else ops.last_corr_launch()
    print('%-28s %9.2f GFLOP %8.3f ms %7.1f TF/s  %s' % (name, flops / 1e9, ms, flops / ms / 1e9, info if args.verbose else ''), flush=True)

enc = [(1, 64, 16384), (64, 128, 4096), (128, 256, 1024), (256, 512, 256), (512, 1024, 64)]
dec = [(2048, 512, 16), (1024, 256, 64), (512, 128, 256), (256, 64, 1024), (128, 1, 4096)]
if args.shape == 'vanilla11':       # the original 11-layer stride-2 SEGAN (train.py:199-205 flags)
    S = 2
    fm = [16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024]
    enc, cin, L = [], 1, 16384
    for c in fm:
        enc.append((cin, c, L)); cin = c; L //= 2
    dec, cin = [], 2 * fm[-1]
    for i, c in enumerate(fm[::-1][1:] + [1]):
        dec.append((cin, c, L)); cin = 2 * c; L *= 2
PADC, PADD = (14, 13) if S == 4 else (ops.conv_pad(K, S)[0], ops.deconv_pad(K, S))
for i, (N, M, L) in enumerate(enc):
    for tag, Nin in (('G', N), ('D', 2 if i == 0 else N)):
        if i > 0 and tag == 'D':
            continue
        name = '%s.enc%d' % (tag if i == 0 else 'GD', i)
        if args.only and args.only not in name:
            continue
        x = torch.randn(B, Nin, L, device=dev)
        w = torch.randn(M, Nin, K, device=dev) * 0.02
        b = torch.zeros(M, device=dev)
        pk = ops.WeightPack()
        fl = 2.0 * B * M * Nin * K * (L // S)
        src = ops.Src(x)
        report(name + ' fwd', fl, timeit(lambda: ops.conv1d_fwd(src, w, b, S, pack=pk)))
        da = torch.randn(B, M, L // S, device=dev)
        report(name + ' dgrad', fl, timeit(lambda: ops.conv1d_dgrad(da, w, L, S, roll=3, pack=pk)))
        dw = torch.zeros_like(w)
        report(name + ' wgrad', fl, timeit(lambda: ops.wgrad(ops.Src(da), src, dw, K, S, PADC, ops.PAD_REFLECT)))
for i, (M, N, Ls) in enumerate(dec):
    name = 'G.dec%d' % i
    if args.only and args.only not in name:
        continue
    x = torch.randn(B, M, Ls, device=dev)
    w = torch.randn(M, N, K, device=dev) * 0.02
    b = torch.zeros(N, device=dev)
    pk = ops.WeightPack()
    fl = 2.0 * B * M * N * K * Ls
    src = ops.Src(x)
    report(name + ' fwd', fl, timeit(lambda: ops.deconv1d_fwd(src, w, b, S, pack=pk)))
    dy = torch.randn(B, N, S * Ls, device=dev)
    report(name + ' dgrad', fl, timeit(lambda: ops.deconv1d_dgrad(dy, w, S, 0, pack=pk)))
    dw = torch.zeros_like(w)
    report(name + ' wgrad', fl, timeit(lambda: ops.wgrad(src, ops.Src(dy), dw, K, S, PADD, ops.PAD_ZERO)))
# packing
w = torch.randn(2048, 512, 31, device=dev)
pk = ops.WeightPack()
def repack():
    ops.bump_weights_epoch(); pk.f(w, S); pk.t(w, S, PADD)
report('pack f+t 2048x512x31', 2048 * 512 * 31 * 4.0 * 4, timeit(repack))
tot_ms = sum(r[2] for r in rows[:-1]); tot_fl = sum(r[1] for r in rows[:-1])
print('TOTAL %.1f GFLOP %.2f ms -> %.1f TF/s' % (tot_fl, tot_ms, tot_fl / tot_ms))
