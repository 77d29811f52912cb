"""Effective shader clock while the contraction kernels run (DVFS under the real load).

A one-wave probe kernel (scripts/clock_probe.hip) samples s_memtime / s_memrealtime on a side
stream while the layer kernel under test is launched back to back on the main stream; the
median clock over the middle half of the loaded window is reported together with the rate the
kernel reached in that window.  usage: python scripts/clock_under_load.py > profiles/rNN_clock_under_load.json
"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segan_pytorch_amd import ops

so = os.path.join(ROOT, 'scripts', 'libclockprobe.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so,
                           os.path.join(ROOT, 'scripts', 'clock_probe.hip')])
lib = ctypes.CDLL(so)
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = 'cuda'
B, K, S = 300, 31, 4
PREC = ops.get_precision()          # SEGAN_PRECISION=bf16: the bf16 contractions against the 2.5 PF dense peak
PEAK = 2500.0 if PREC == 'bf16' else 2500.0 / 6 if PREC == 'bf16x3' else 157.3
side = torch.cuda.Stream()


def probe(fn, reps, what, flops):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms1 = e0.elapsed_time(e1)
    n = int((reps * ms1 + 4.0) / 0.02)            # one sample every 20 us, 2 ms idle either side
    buf = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    with torch.cuda.stream(side):
        lib.clock_probe_launch(ctypes.c_void_p(buf.data_ptr()), n, 2000, ctypes.c_void_p(side.cuda_stream))
    import time; time.sleep(0.002)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    b = buf.cpu().view(-1, 2).double()
    dc, dr = b[1:, 0] - b[:-1, 0], b[1:, 1] - b[:-1, 1]
    mhz = (dc / dr * 100.0)
    lo, hi = int(0.3 * len(mhz)), int(0.7 * len(mhz))
    return {'kernel': what, 'ms': ms, 'tflops': flops / ms / 1e9,
            'clock_mhz_loaded_median': float(mhz[lo:hi].median()),
            'clock_mhz_loaded_min': float(mhz[lo:hi].min()), 'clock_mhz_loaded_max': float(mhz[lo:hi].max()),
            'clock_mhz_idle_head': float(mhz[:20].median()),
            'mfma_peak_at_loaded_clock_tflops': PEAK * float(mhz[lo:hi].median()) / 2400.0,
            'frac_of_nominal_peak': flops / ms / 1e9 / PEAK,
            'frac_of_peak_at_loaded_clock': flops / ms / 1e9 / (PEAK * float(mhz[lo:hi].median()) / 2400.0)}


rows = []
for name, (N, M, L) in (('enc2', (128, 256, 1024)), ('enc4', (512, 1024, 64))):
    x = torch.randn(B, N, L, device=dev); w = torch.randn(M, N, K, device=dev) * 0.02
    b = torch.zeros(M, device=dev); pk = ops.WeightPack(); src = ops.Src(x)
    da = torch.randn(B, M, L // S, device=dev); dw = torch.zeros_like(w)
    fl = 2.0 * B * M * N * K * (L // S)
    rows.append(probe(lambda: ops.conv1d_fwd(src, w, b, S, pack=pk), 20, name + ' conv fwd (F form)', fl))
    rows.append(probe(lambda: ops.conv1d_dgrad(da, w, L, S, pack=pk), 20, name + ' conv dgrad (T form)', fl))
    rows.append(probe(lambda: ops.wgrad(ops.Src(da), src, dw, K, S, 14, ops.PAD_REFLECT), 20, name + ' wgrad (W form)', fl))
M, N, Ls = 512, 128, 256
x = torch.randn(B, M, Ls, device=dev); w = torch.randn(M, N, K, device=dev) * 0.02
b = torch.zeros(N, device=dev); pk = ops.WeightPack(); src = ops.Src(x)
rows.append(probe(lambda: ops.deconv1d_fwd(src, w, b, S, pack=pk), 12, 'dec2 deconv fwd (T form)', 2.0 * B * M * N * K * Ls))
print(json.dumps({'note': 'shader clock sampled by a one-wave probe kernel on a side stream while the kernel under '
                          'test runs back to back (B=300, precision {}; bf16 rows include the packing pass of the entry '
                          'point, so the clock is that of the contraction kernel but the rate is the entry point\'s)'.format(PREC),
                  'precision': PREC, 'nominal_peak_tflops': PEAK, 'rows': rows}, indent=1))
