#!/usr/bin/env python
"""Benchmark of the SEGAN+ GAN training step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full LSGAN step of the reference's ``SEGAN.train`` inner loop
(segan/models/model.py:292-321: G fwd, D fwd real+fake, D bwd, D step, D fwd on fake,
G bwd + L1, G step) on a batch of 300 synthetic 16384-sample noisy/clean pairs PER GPU
(BASELINE.json configs[1]: SEGAN+ default net, k31, batch 300, fp32), inputs resident
in HBM before the timed region, no logging syncs inside it.  With N > 1 it is launched
by torch.distributed.run (one rank per GPU); the batch is sharded 300/GPU (weak
scaling) and gradients are averaged with one RCCL all-reduce per network per step.

Prints ONE JSON line (rank 0):
  value       whole-job 16384-sample chunks/s
  roofline    the dominant kernel family (corr_kernel: every conv/deconv forward and data
              gradient): ALGORITHMIC flops of its launches / their summed duration,
              measured live with HIP events on the launch stream, vs the 157.3 TF/s
              fp32-MFMA peak (guides/MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/segan_oracle.py, a port of the reference's path)
              timed on this host's cores on a bounded sample, rank 0 at N=1 only
"""
import argparse
import json
import os
import random
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_CHUNK = 37.96      # 2*(3*3144.94 + 9*1060.67) MMAC
MB_PER_CHUNK = 80.0          # SURVEY.md 8(d) algorithmic HBM bytes (fp32)
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def default_opts(save_path='/tmp/segan_bench_ckpt'):
    """ckpt_segan+/train.opts of the reference (the SEGAN+ release configuration)."""
    return dict(save_path=save_path, preemph=0.95, reg_loss='l1_loss', batch_size=300,
                epoch=1, opt='rmsprop', g_lr=5e-5, d_lr=5e-5, l1_weight=100, l1_dec_step=1e-5,
                l1_dec_epoch=100, skip_merge='concat', skip_type='alpha', skip_init='one',
                skip_kwidth=11, gkwidth=31, genc_fmaps=[64, 128, 256, 512, 1024],
                genc_poolings=[4, 4, 4, 4, 4], z_dim=1024, gdec_fmaps=None, gdec_poolings=None,
                gdec_kwidth=None, no_z=False, no_skip=False, denc_fmaps=[64, 128, 256, 512, 1024],
                dpool_type='none', dpool_slen=16, dkwidth=None, denc_poolings=[4, 4, 4, 4, 4],
                dnorm_type='bnorm', phase_shift=5, sinc_conv=False, bias=True, seed=111)


class KernelTimer(object):
    """Brackets every launch of the contraction entry points with HIP events on torch's
    current stream (the stream the kernels are launched on) and books the algorithmic
    FLOPs of the call."""

    CORR = ('conv1d_fwd', 'conv1d_dgrad', 'deconv1d_fwd', 'deconv1d_dgrad')

    def __init__(self):
        self.records = []     # (family, flops, ev0, ev1)
        self._saved = {}

    @staticmethod
    def _flops(name, args, kwargs, out):
        # 2 * B * Cout * Cin * K * Lout-equivalents
        if name == 'conv1d_fwd':
            src, w = args[0], args[1]
            return 2.0 * src.B * w.shape[0] * w.shape[1] * w.shape[2] * (src.L // args[3])
        if name == 'conv1d_dgrad':
            da, w = args[0], args[1]
            return 2.0 * da.shape[0] * w.shape[0] * w.shape[1] * w.shape[2] * da.shape[2]
        if name == 'deconv1d_fwd':
            src, w = args[0], args[1]
            return 2.0 * src.B * w.shape[0] * w.shape[1] * w.shape[2] * src.L
        if name == 'deconv1d_dgrad':
            dy, w, S = args[0], args[1], args[2]
            M0 = args[3] if len(args) > 3 else kwargs.get('M0', 0)
            rows = w.shape[0] - (M0 if kwargs.get('need0', True) is False else 0)
            return 2.0 * dy.shape[0] * rows * w.shape[1] * w.shape[2] * (dy.shape[2] // S)
        if name == 'wgrad':
            lo, hi, dw = args[0], args[1], args[2]
            return 2.0 * lo.B * dw.shape[0] * dw.shape[1] * dw.shape[2] * lo.L
        return 0.0

    def install(self):
        from segan_pytorch_amd import ops
        for name in self.CORR + ('wgrad',):
            fn = getattr(ops, name)
            self._saved[name] = fn
            fam = 'wgrad' if name == 'wgrad' else 'corr'

            def wrapped(*a, _fn=fn, _name=name, _fam=fam, **k):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _fn(*a, **k)
                e1.record()
                self.records.append((_fam, self._flops(_name, a, k, out), e0, e1))
                return out
            setattr(ops, name, wrapped)

    def uninstall(self):
        from segan_pytorch_amd import ops
        for name, fn in self._saved.items():
            setattr(ops, name, fn)
        self._saved = {}

    def summary(self):
        out = {}
        for fam in ('corr', 'wgrad'):
            rs = [r for r in self.records if r[0] == fam]
            if not rs:
                continue
            ms = sum(r[2].elapsed_time(r[3]) for r in rs)
            fl = sum(r[1] for r in rs)
            out[fam] = dict(launches=len(rs), total_ms=ms, avg_us=1e3 * ms / len(rs),
                            tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                            flops_per_launch=fl / len(rs))
        return out


def cpu_baseline(B=48):
    """Time the CPU oracle's GAN step (oracle/segan_oracle.py: the reference's path restated
    on torch CPU ops) on this host's cores: one warm-up step + one timed step at batch B with
    oneDNN off (the numerically trustworthy setting) and the same again with
    oneDNN on (what a stock reference run would use); the FASTER of the two is reported."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import segan_oracle as O
    from segan_pytorch_amd.datasets import synthetic_pairs
    from segan_pytorch_amd.models import SEGAN, WSEGAN
    opts = default_opts()
    random.seed(111); np.random.seed(111); torch.manual_seed(111)
    m = SEGAN(SimpleNamespace(**opts))
    gsd0 = {k: v.detach() for k, v in m.G.state_dict().items()}
    dsd0 = {k: v.detach() for k, v in m.D.state_dict().items()}
    clean, noisy = synthetic_pairs(B, 16384, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(B, 1024, 16)
    rolls = [[1, -2, 3, -4, 5]] * 3
    st = opts['genc_poolings']
    results = {}
    for mode in (False, True):
        torch.backends.mkldnn.enabled = mode
        gsd, dsd, g_sq, d_sq = gsd0, dsd0, None, None
        per = None
        for i in range(2):
            t0 = time.perf_counter()
            res = O.gan_step(gsd, dsd, clean, noisy, z, rolls, st, 100.0, 5e-5, g_sq=g_sq,
                             d_sq=d_sq)
            per = time.perf_counter() - t0
            gsd, dsd, g_sq, d_sq = res['G'], res['D'], res['g_sq'], res['d_sq']
        results['onednn_on' if mode else 'onednn_off'] = per
    torch.backends.mkldnn.enabled = False
    best = min(results, key=results.get)
    per = results[best]
    return dict(value=B / per, unit='chunks/s', cores=torch.get_num_threads(), kind='port',
                sample='oracle GAN step (SEGAN+ default net, fp32) at batch {}: 1 warm-up + 1 timed '
                       'step per setting; {:.2f} s/step oneDNN off, {:.2f} s/step oneDNN on; '
                       'reported = {}'.format(B, results['onednn_off'], results['onednn_on'], best))


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE in separate passes over this
    same command; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md)."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')
    if not os.path.exists(path):
        return None
    try:
        ks = json.load(open(path))['kernels']
        n = f = w = 0.0
        for name, v in ks.items():
            if kernel_prefix in name:
                n += v['launches']
                f += v['launches'] * v['fetch_kb_avg']
                w += v['launches'] * v['write_kb_avg']
        if n == 0:
            return None
        return (2.0 * f + w) * 1024.0 / n
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=300, help='chunks per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--no-modes', action='store_true', help='skip the bf16x3 / bf16 side measurements')
    ap.add_argument('--wsegan', action='store_true',
                    help='time the WSEGAN step of BASELINE config 4 (--wsegan --misalign_pair) instead '
                         'of the SEGAN+ step; a side measurement, not the headline metric')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16x3', 'bf16'],
                    help='forward/data-gradient contraction precision (default: exact fp32, the '
                         'BASELINE configuration; bf16x3 = exact 3-way bf16 split of fp32 operands; '
                         'bf16 = BASELINE config 5)')
    args = ap.parse_args()

    from segan_pytorch_amd import distributed as sdist
    from segan_pytorch_amd import losses
    from segan_pytorch_amd.datasets import synthetic_pairs
    from segan_pytorch_amd.models import SEGAN, WSEGAN

    from segan_pytorch_amd import ops as _ops
    _ops.set_precision(args.precision)
    rank, world, local = sdist.init_from_env()
    if world != max(1, args.gpus) and world > 1:
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
    dev = torch.device('cuda', local if world > 1 else 0)
    torch.cuda.set_device(dev)

    opts = default_opts()
    random.seed(111); np.random.seed(111); torch.manual_seed(111)
    if args.wsegan:
        opts.update(dict(misalign_pair=True, interf_pair=False, pow_weight=0.001, vanilla_gan=False,
                         n_fft=2048))
        model = WSEGAN(SimpleNamespace(**opts)).to(dev)
    else:
        model = SEGAN(SimpleNamespace(**opts)).to(dev)
    o = SimpleNamespace(**opts)
    Gopt, Dopt = model.build_optimizers(o)
    sdist.broadcast_params(model.G)
    sdist.broadcast_params(model.D)
    model.G.train()
    model.D.train()
    criterion = losses.MSELoss()
    B = args.batch
    clean, noisy = synthetic_pairs(B, 16384, seed=rank, device=dev)
    clean, noisy = clean.unsqueeze(1).contiguous(), noisy.unsqueeze(1).contiguous()
    random.seed(1000 + rank)
    zgen = torch.Generator(device=dev).manual_seed(rank)

    names = ['utt_additive_{}'.format(i) if i % 2 == 0 else 'utt_{}'.format(i) for i in range(B)]

    def one_step():
        z = torch.randn(B, 1024, 16, device=dev, generator=zgen)
        if args.wsegan:
            return model.wgan_step(names, clean, noisy, Gopt, Dopt, 100.0, z=z)
        return model.gan_step(clean, noisy, Gopt, Dopt, criterion, 100.0, z=z)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    timer = None
    if not args.no_kernel_timer:
        timer = KernelTimer()
        timer.install()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses_out = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if timer is not None:
        timer.uninstall()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    finite = all(bool(torch.isfinite(x)) for x in losses_out)

    # Same step with the contractions on the bf16 matrix cores, reported BESIDE the fp32
    # headline (never as `value`): 'bf16x3' = exact 3-way split of the fp32 operands,
    # 'bf16' = BASELINE config 5.  Every rank runs the same steps (the collectives match).
    modes = {}
    if args.precision == 'fp32' and not args.no_modes:
        for prec in ('bf16x3', 'bf16'):
            try:
                _ops.set_precision(prec)
                for _ in range(2):
                    one_step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    lo = one_step()
                barrier()
                dm = time.perf_counter() - t1
                if world > 1:
                    t = torch.tensor([dm], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    dm = float(t.item())
                modes[prec] = {'value': B * world * args.steps / dm, 'unit': 'chunks/s',
                               'ms_per_step': 1e3 * dm / args.steps,
                               'losses_finite': all(bool(torch.isfinite(x)) for x in lo)}
            except Exception as e:      # a side measurement must never cost the headline line
                modes[prec] = {'error': repr(e)}
                break
            finally:
                _ops.set_precision('fp32')

    if rank == 0:
        chunks = B * world * args.steps
        value = chunks / dt
        ms = 1e3 * dt / args.steps
        line = {
            'metric': '16384-sample waveform chunks/sec (GAN step)', 'value': value,
            'unit': 'chunks/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16x3': 'f32 operands as 3 bf16 planes (6 products), f32 accumulate',
                      'bf16': 'bf16 operands, f32 accumulate'}[args.precision], 'data': 'synthetic',
            'config': {'workload': ('WSEGAN step with --misalign_pair (model.py:577-669; BASELINE '
                                    'config 4), same nets, batch {} x 16384-sample chunks per GPU'
                                    if args.wsegan else
                                    'SEGAN+ default G+D (5+5 layers, k31, stride 4, z 1024x16), '
                                    'batch {} x 16384-sample chunks per GPU, full GAN step '
                                    '(model.py:292-321), RMSprop, fp32').format(B),
                       'global_batch': B * world, 'parallelism': 'dp{}'.format(world)},
            'losses_finite': finite,
            'precision': args.precision,
            'step_tflops': (44.33 if args.wsegan else GFLOP_PER_CHUNK) * value / 1e3,
            'step_frac_of_f32_mfma_peak': (44.33 if args.wsegan else GFLOP_PER_CHUNK) * value / 1e3 / PEAK_F32_MFMA_TF / world,
            'step_hbm_gbs_algorithmic': MB_PER_CHUNK * value / 1e3 / world,
        }
        if timer is not None:
            s = timer.summary()
            c = s.get('corr')
            if c:
                line['roofline'] = {
                    'bound': 'mfma', 'kernel': 'corr_kernel (conv/deconv forward + data gradient)',
                    'achieved': c['tflops'], 'peak': PEAK_F32_MFMA_TF, 'unit': 'TFLOP/s',
                    'frac': c['tflops'] / PEAK_F32_MFMA_TF, 'traffic': pmc_traffic('corr_kernel'),
                    'traffic_unit': 'HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, '
                                    'profiles/r01_pmc_hbm_traffic.json)',
                    'avg_launch_us': c['avg_us'], 'launches': c['launches'],
                    'gflop_per_launch': c['flops_per_launch'] / 1e9,
                    'share_of_step_time': c['total_ms'] / (1e3 * dt)}
            if 'wgrad' in s:
                w = s['wgrad']
                line['roofline_wgrad'] = {'bound': 'mfma', 'kernel': 'wgrad_kernel',
                                          'achieved': w['tflops'], 'peak': PEAK_F32_MFMA_TF,
                                          'unit': 'TFLOP/s', 'frac': w['tflops'] / PEAK_F32_MFMA_TF,
                                          'avg_launch_us': w['avg_us'], 'launches': w['launches'],
                                          'share_of_step_time': w['total_ms'] / (1e3 * dt)}
        if modes:
            modes['note'] = ('same workload, contractions on the bf16 MFMA: bf16x3 = fp32 operands '
                             'split exactly into 3 bf16 planes, 6 partial products, fp32 accumulate '
                             '(parity tolerance 5e-5, tests/test_gpu_kernels.py); bf16 = BASELINE '
                             'config 5 (tolerance 2e-2).  `value` above is the exact-fp32 run.')
            line['other_precisions'] = modes
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline()
                line['speedup_vs_cpu_baseline'] = value / line['cpu_baseline']['value']
            except Exception as e:  # the bench line must still be printed
                line['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
