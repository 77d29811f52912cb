#!/usr/bin/env python
"""Benchmark of the SEGAN+ GAN training step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full LSGAN step of the reference's ``SEGAN.train`` inner loop
(segan/models/model.py:292-321: G fwd, D fwd real+fake, D bwd, D step, D fwd on fake,
G bwd + L1, G step) on a batch of 300 synthetic 16384-sample noisy/clean pairs PER GPU
(BASELINE.json configs[1]: SEGAN+ default net, k31, batch 300, fp32), inputs resident
in HBM before the timed region, no logging syncs inside it.  With N > 1 there is one rank per
GPU: when the driver has not launched the ranks itself (WORLD_SIZE unset) the script re-executes
under ``python -m torch.distributed.run --nproc-per-node N``; the batch is sharded 300/GPU
(weak scaling) and gradients are averaged by RCCL all-reduces on the flat gradient arenas.

Prints ONE JSON line (rank 0):
  value       whole-job 16384-sample chunks/s
  roofline    the dominant kernel family (corr_kernel: every conv/deconv forward and data
              gradient): ALGORITHMIC flops of its launches / their summed duration,
              measured live with HIP events on the launch stream in three of the K timed steps
              (KernelTimer: every launch of those steps), vs the 157.3 TF/s fp32-MFMA peak
              (guides/MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/segan_oracle.py, a port of the reference's path)
              timed on this host's cores at the metric's batch (300), rank 0 at N=1 only
  parity      BASELINE's second metric: the HIP step against that same oracle step (same
              weights, inputs, z and phase shifts): generator-output MSE / max-abs, losses and
              gradients
  other_precisions   the same step with the contractions on the bf16 matrix cores (bf16x3, bf16 =
              BASELINE config 5), timed beside the fp32 headline, never as `value`
  other_workloads    BASELINE's other fp32 configurations, timed like the headline: `wsegan` (config 4:
              --wsegan --misalign_pair) and `vanilla11` (the 11-layer stride-2 shape config 2 words),
              each with its own roofline blocks, executed FLOPs and PMC traffic
  host        what the launch path costs: host_enqueue_ms_per_step (python + ctypes + HIP launches of
              one step, device idle at its start), gpu_ms_per_step_unstarved (HIP-event time of a step
              whose launches were all queued behind a blocker), gpu_idle_ms_per_step = their gap to the
              timed step
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
PEAK_BF16_MFMA_TF = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md
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


VANILLA11 = dict(genc_fmaps=[16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024],
                 denc_fmaps=[16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024],
                 genc_poolings=[2] * 11, denc_poolings=[2] * 11, dpool_slen=8)


def gflop_per_chunk(o, T=16384, wsegan=False, executed=False):
    """Algorithmic FLOPs of one GAN step per chunk, SURVEY.md 8(d) accounting: 2 * (3 * G forward
    MACs + 9 * D forward MACs) (12 * D with the WSEGAN misalign pair), from the layer shapes.
    `executed`: what this engine runs — the D weight gradients of the generator phase (which the
    reference computes and discards at its next Dopt.zero_grad()) are not computed: one D
    forward-equivalent less."""
    K = o['gkwidth']
    g, cin, L = 0, 1, T
    for c, s in zip(o['genc_fmaps'], o['genc_poolings']):
        L //= s
        g += c * cin * K * L
        cin = c
    dec = o['genc_fmaps'][::-1][1:] + [1]
    cin = o['genc_fmaps'][-1] + (0 if o['no_z'] else o['z_dim'])
    for i, (c, s) in enumerate(zip(dec, o['genc_poolings'][::-1])):
        if i > 0 and not o['no_skip']:
            cin *= 2
        g += cin * c * K * L
        L *= s
        cin = c
    d, cin, L = 0, 2, T
    for c, s in zip(o['denc_fmaps'], o['denc_poolings']):
        L //= s
        d += c * cin * K * L
        cin = c
    d += cin * L * 256 + 256 * 128 + 128
    return 2.0 * (3 * g + ((12 if wsegan else 9) - (1 if executed else 0)) * d) / 1e9


def z_lookahead_ok(wsegan):
    """Is the next step's z drawn one step ahead on a host thread in this workload's training loop?
    SEGAN.train: yes (models/model.py).  WSEGAN.train: yes since round 6 on the full-rate input path
    (train.py --pcm_shard: PCMShardLoader.sample() draws from a private generator, so nothing but the z
    draws takes from torch's global CPU generator between two steps); with a plain DataLoader the
    reference's per-step next(iter(dloader)) reseeds from the global generator between two z draws
    (model.py:526-535, generator.py:197) and WSEGAN.train keeps the synchronous draw.  The bench's
    resident batch stands for the full-rate path."""
    return True


def sample_steps(steps):
    """The timed steps whose contraction launches are bracketed with HIP events: the first, the middle
    and the last one."""
    return sorted({0, steps // 2, steps - 1}) if steps > 0 else []


class KernelTimer(object):
    """Brackets the launches of the contraction entry points with HIP events on torch's
    current stream (the stream the kernels are launched on) and books the algorithmic
    FLOPs of the call.

    The instrumentation must not change what it measures (round 6: with two freshly created timing
    events around EVERY contraction call the 11-layer shape — 151 calls per step — ran 78.6 instead
    of 68.1 ms per step in a fresh process, the SEGAN+ step 86.0 instead of 85.4; on a slower host
    more: the "host-bound on some boxes" of the round-5 review was this).  So: (1) only the steps of
    `sample_steps` are instrumented — three of the K timed steps, every launch of those —, the others
    run the plain entry points; (2) the events are created BEFORE the timed region (`prepare`: the
    last warm-up step runs instrumented, which creates one step's worth of HIP events; two more
    steps' worth are created and recorded once) and taken from that pool inside it."""

    CORR = ('conv1d_fwd', 'conv1d_dgrad', 'deconv1d_fwd', 'deconv1d_dgrad')

    def __init__(self):
        self.records = []     # (family, flops, ev0, ev1)
        self._saved = {}
        self.active = True
        self.sampled = 0      # instrumented steps so far
        self._pool, self._next = [], 0

    def _event(self):
        if self._next < len(self._pool):
            e = self._pool[self._next]
        else:
            e = torch.cuda.Event(enable_timing=True)
            self._pool.append(e)
        self._next += 1
        return e

    def begin_step(self, sampled):
        self.active = bool(sampled)
        if sampled:
            self.sampled += 1

    def prepare(self, one_step, nsteps):
        """Run ONE instrumented (untimed) step to learn how many events a step takes and to create
        them, create the events of `nsteps` - 1 further steps (torch creates the HIP event on the first
        record), then forget the dry step's records."""
        self.begin_step(True)
        one_step()
        per_step = self._next
        for _ in range(per_step * max(nsteps - 1, 0)):
            self._event().record()
        self.records, self._next, self.sampled, self.active = [], 0, 0, False
        return per_step

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
        if name == 'gemm':      # gemm(A, sam, sak, Bm, sbk, sbn, C, M, N, K, overwrite)
            return 2.0 * args[7] * args[8] * args[9]
        return 0.0

    def install(self):
        from segan_pytorch_amd import ops
        for name in self.CORR + ('wgrad', 'gemm'):
            fn = getattr(ops, name)
            self._saved[name] = fn
            fam = name if name in ('wgrad', 'gemm') else 'corr'

            def wrapped(*a, _fn=fn, _name=name, _fam=fam, **k):
                if not self.active:
                    return _fn(*a, **k)
                e0, e1 = self._event(), self._event()
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

    def booked_flops_per_step(self):
        """FLOPs the launches of one instrumented step were booked with: the matrix work the engine
        EXECUTES per step (conv / deconv forward, data and weight gradients, the dense-head and STFT
        GEMMs)."""
        if self._done is not None:
            return self._done[1]
        return sum(r[1] for r in self.records) / max(self.sampled, 1)

    _done = None

    def finish(self):
        """Read the events (the device must be idle: call after the closing barrier), keep the figures
        and DROP the events: hundreds of live timing events are destroyed here, outside any timed
        region, not whenever the collector finds the timer."""
        self._done = (self.summary(), self.booked_flops_per_step())
        self.records, self._pool, self._next = [], [], 0
        return self._done[0]

    def summary(self):
        if self._done is not None:
            return self._done[0]
        out = {}
        for fam in ('corr', 'wgrad', 'gemm'):
            rs = [r for r in self.records if r[0] == fam]
            if not rs:
                continue
            ms = sum(r[2].elapsed_time(r[3]) for r in rs)
            fl = sum(r[1] for r in rs)
            n = max(self.sampled, 1)
            out[fam] = dict(launches=len(rs), launches_per_step=len(rs) / n, sampled_steps=self.sampled,
                            total_ms=ms, ms_per_step=ms / n, avg_us=1e3 * ms / len(rs),
                            tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                            flops_per_launch=fl / len(rs))
        return out


def run_timed(one_step, steps, warmup, barrier, timer=None, on_timed_start=None):
    """`warmup` untimed steps, then EXACTLY `steps` steps timed between two barriers (barrier +
    device synchronisation on both sides).  With a KernelTimer the last warm-up step creates its
    events and the steps of `sample_steps` are instrumented.  Returns (seconds, last step's output)."""
    import gc
    samp = sample_steps(steps) if timer is not None else []
    for i in range(warmup):
        if timer is not None and i == warmup - 1:
            timer.prepare(one_step, len(samp))
        else:
            one_step()
    # no cyclic collection inside the timed region (a generation-2 pass over a process that holds two
    # networks and thousands of tensors is tens of milliseconds of host time at a random launch)
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    if on_timed_start is not None:
        on_timed_start()        # e.g. the data-parallel wait statistics: of the timed steps only
    try:
        barrier()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            if timer is not None:
                timer.begin_step(i in samp)
            out = one_step()
        barrier()
        dt = time.perf_counter() - t0
    finally:
        if gc_was:
            gc.enable()
    if timer is not None:
        timer.active = False
        timer.finish()
    return dt, out


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def _lrel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


def hip_step_parity(ref, opts, gsd0, dsd0, clean, noisy, z, rolls, dev, precision, deterministic,
                    accumulation='plain'):
    """One HIP GAN step from (gsd0, dsd0) on (clean, noisy, z, rolls) in the given contraction
    precision / reduction mode against the oracle step `ref` from the same state: generator
    output, the four losses, gradients per tensor (relative L2, worst tensor).  The generator
    phase runs through the ORACLE's post-step discriminator (RMSprop's first step is
    ill-conditioned where |g| is at roundoff level: DESIGN.md section 6)."""
    from segan_pytorch_amd import losses, ops
    from segan_pytorch_amd.models import SEGAN
    old_d, old_p, old_a = ops.get_deterministic(), ops.get_precision(), ops.get_accumulation()
    ops.set_deterministic(deterministic)
    ops.set_precision(precision)
    ops.set_accumulation(accumulation)
    try:
        mm = SEGAN(SimpleNamespace(**opts))
        mm.G.load_state_dict(gsd0)
        mm.D.load_state_dict(dsd0)
        mm = mm.to(dev)
        Gopt, Dopt = mm.build_optimizers(SimpleNamespace(**opts))
        mm.G.train(); mm.D.train()
        it = iter(rolls)
        mm.D.draw_rolls = lambda: list(next(it))
        crit = losses.MSELoss()
        cg, ng, zg = clean.to(dev), noisy.to(dev), z.to(dev)
        Genh, d_real, d_fake = mm.d_phase(cg, ng, Dopt, crit, z=zg)
        y = Genh.detach().cpu().double()
        yr = ref['Genh'].double()
        dn = dict(mm.D.named_parameters())
        d_grad = max(_rel(dn[k].grad, g) for k, g in ref['d_grads'].items()
                     if not k.endswith('conv.bias'))
        mm.D.load_state_dict({k: ref['D'].get(k, v) for k, v in dsd0.items()})
        ops.bump_weights_epoch()
        g_adv, g_l1 = mm.g_phase(Genh, cg, ng, Gopt, crit, 100.0)
        torch.cuda.synchronize()
        gn = dict(mm.G.named_parameters())
        g_grad = max(_rel(gn[k].grad, g) for k, g in ref['g_grads'].items())
        out = {
            'batch': int(clean.size(0)), 'precision': precision, 'accumulation': accumulation,
            'reduction_mode': 'deterministic (fixed-order reductions)' if deterministic else
                              'default (fp32 atomics in the weight-gradient / dense-head splits: the timed mode)',
            'g_mse': ((y - yr) ** 2).mean().item(), 'g_max_abs': (y - yr).abs().max().item(),
            'd_real_loss_rel': _lrel(d_real, ref['d_real_loss']),
            'd_fake_loss_rel': _lrel(d_fake, ref['d_fake_loss']),
            'g_adv_loss_rel': _lrel(g_adv, ref['g_adv_loss']),
            'g_l1_loss_rel': _lrel(g_l1, ref['g_l1_loss']),
            'd_grad_rel_l2_worst_tensor': d_grad, 'g_grad_rel_l2_worst_tensor': g_grad}
        del mm, Gopt, Dopt
        return out
    finally:
        ops.set_deterministic(old_d)
        ops.set_precision(old_p)
        ops.set_accumulation(old_a)


PARITY_NOTE = ('HIP step vs the CPU oracle step timed above, same weights / inputs / z / phase shifts; '
               'generator phase through the oracle\'s post-step D; gradient figures are relative L2 per '
               'tensor (two dozen ReLU-gate flips at fp32 roundoff bound them: with the gates aligned the distance '
               'is 5e-6, tests/test_gpu_kernels.py::test_discriminator_gradients_with_aligned_gates); `parity` = fp32 in the deterministic mode '
               '(bit-reproducible), `parity_default_mode` = fp32 in the timed (atomics) mode, '
               'other_precisions.*.parity = the bf16x3 / bf16 contractions in the timed mode; '
               '`parity_blocked_accumulation` = fp32 with ops.set_accumulation(\'blocked\') '
               '(SEGAN_PREC_FP32_BLOCKED, opt-in: timed as ms_per_step_blocked_accumulation)')


def port_vs_reference():
    """The newest committed measurement of the oracle ("port") against the reference's literal
    SEGAN.train on one host (oracle/time_ref_vs_port.py, build container: needs /root/reference)."""
    path = _profile('ref_vs_port_cpu.json')
    try:
        d = json.load(open(path))
        r = d['rows']
        return ('{}: port / reference time per step {:.2f} (oneDNN on), {:.2f} (off) at batch {} on {} '
                'threads in the build container'.format(os.path.relpath(path, ROOT),
                                                        r['onednn_on']['port_over_reference'],
                                                        r['onednn_off']['port_over_reference'], d['batch'],
                                                        d['threads']))
    except Exception:
        return None


def cpu_baseline(B=300, steps=3, dev=None, modes=('bf16x3', 'bf16'), budget_s=150.0):
    """Time the CPU oracle's GAN step (oracle/segan_oracle.py: the reference's path restated on
    torch CPU ops; /root/reference does not exist on the GPU box, hence kind = 'port';
    profiles/rNN_ref_vs_port_cpu.json: timed back to back with the reference's literal SEGAN.train
    in the build container the port takes 0.90 - 1.00 (round 4) / 0.92 - 0.97 (round 6) of the
    reference's time per step) on this host's cores at the metric's batch size, SURVEY.md 8d's protocol: per oneDNN
    setting one warm-up step at batch 8 (thread pools, allocator), then `steps` >= 3 timed steps
    at batch B; a setting's figure is the MEAN OF ITS STEPS >= 2, the FASTER setting is reported,
    min and mean both stated.  To bound the run (a step is 35-75 s on the boxes seen) the
    oneDNN-off setting — the numerically trustworthy one, SURVEY.md 0.4b, slower on every host
    seen — runs ONE step first (it is the parity reference below) and only gets its remaining
    steps if that one step is not already slower than the oneDNN-on mean.  `budget_s`: on a busy
    shared host (50 - 150 s per step were seen) a setting stops at two timed steps once the baseline
    has run that long; the `sample` string says how many steps ran.

    The oneDNN-off step doubles as the parity reference: the HIP model takes the
    same step from the same weights / inputs / z / phase shifts — in fp32 in the deterministic
    and in the default (timed) reduction mode, and with the bf16x3 / bf16 contractions — and the
    differences are returned as a dict of parity blocks (hip_step_parity)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import segan_oracle as O
    from segan_pytorch_amd.datasets import synthetic_pairs
    from segan_pytorch_amd.models import SEGAN
    steps = max(3, int(steps))
    opts = default_opts()
    random.seed(111); np.random.seed(111); torch.manual_seed(111)
    m = SEGAN(SimpleNamespace(**opts))
    gsd0 = {k: v.detach().clone() for k, v in m.G.state_dict().items()}
    dsd0 = {k: v.detach().clone() for k, v in m.D.state_dict().items()}
    clean, noisy = synthetic_pairs(B, 16384, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(B, 1024, 16, generator=torch.Generator().manual_seed(0))
    rolls = [[1, -2, 3, -4, 5], [-3, 2, -1, 5, 4], [2, -5, 1, -1, -4]]
    st = opts['genc_poolings']
    results, ref = {'onednn_off': [], 'onednn_on': []}, None
    state = {}

    t_begin = time.perf_counter()

    def run(mode, n):
        nonlocal ref
        key = 'onednn_on' if mode else 'onednn_off'
        torch.backends.mkldnn.enabled = mode
        if key not in state:
            O.gan_step(gsd0, dsd0, clean[:8], noisy[:8], z[:8], rolls, st, 100.0, 5e-5)   # warm-up
            state[key] = (gsd0, dsd0, None, None)
        gsd, dsd, g_sq, d_sq = state[key]
        for _ in range(n):
            # a busy host (load averages of 30+ on these shared boxes: 50 - 150 s per oracle step instead
            # of 35 - 70) must not turn the default run into ten minutes: once the budget is spent a
            # setting stops at two steps (the protocol's figure is the mean of the steps >= 2)
            if len(results[key]) >= 2 and time.perf_counter() - t_begin > budget_s:
                break
            t0 = time.perf_counter()
            res = O.gan_step(gsd, dsd, clean, noisy, z, rolls, st, 100.0, 5e-5, g_sq=g_sq, d_sq=d_sq)
            results[key].append(time.perf_counter() - t0)
            if not mode and ref is None:
                ref = res
            gsd, dsd, g_sq, d_sq = res['G'], res['D'], res['g_sq'], res['d_sq']
        state[key] = (gsd, dsd, g_sq, d_sq)
        torch.backends.mkldnn.enabled = False

    def figure(ts):          # mean of the steps >= 2 (all of them if fewer ran)
        tail = ts[1:] if len(ts) > 1 else ts
        return sum(tail) / len(tail)

    run(False, 1)
    run(True, steps)
    if results['onednn_off'][0] <= figure(results['onednn_on']) and time.perf_counter() - t_begin < budget_s:
        run(False, steps - 1)
    state.clear()
    fig = {k: figure(v) for k, v in results.items()}
    best = min(fig, key=fig.get)
    out = dict(value=B / fig[best], unit='chunks/s', cores=torch.get_num_threads(),
               nproc=os.cpu_count(), kind='port',
               value_best_step=B / min(results[best]),
               seconds_per_step={k: {'steps': [round(t, 2) for t in v], 'min': round(min(v), 2),
                                     'mean_steps_ge2': round(fig[k], 2)} for k, v in results.items()},
               reported=best,
               port_vs_reference=port_vs_reference(),
               sample='oracle GAN step (SEGAN+ default net, fp32) at batch {}: warm-up at batch 8, then '
                      '{} timed steps with oneDNN on and {} with it off; value = batch / mean of the '
                      'steps >= 2 of the faster setting ({}); value_best_step = batch / its fastest '
                      'step'.format(B, len(results['onednn_on']), len(results['onednn_off']), best))
    parity = {}
    if dev is not None and ref is not None:
        for name, prec, det, acc in (('fp32_deterministic', 'fp32', True, 'plain'),
                                     ('fp32_default', 'fp32', False, 'plain'),
                                     ('fp32_blocked', 'fp32', False, 'blocked')) + \
                tuple((p, p, False, 'plain') for p in modes):
            try:
                parity[name] = hip_step_parity(ref, opts, gsd0, dsd0, clean, noisy, z, rolls, dev,
                                               prec, det, acc)
            except Exception as e:      # a parity leg must never cost the bench line
                parity[name] = {'error': repr(e)}
    return out, parity


def wsegan_parity(opts, B, dev):
    """BASELINE config 4 at its benchmarked batch: ONE step of the oracle's WSEGAN step
    (oracle.wsegan_step: --misalign_pair, LSGAN cost, STFT power loss; oneDNN off) timed on the
    host cores, and the HIP WSEGAN step from the same weights / inputs / z / phase shifts /
    misalign permutation compared with it (fp32, both reduction modes; the generator phase through
    the oracle's post-step D).  Returns (cpu_baseline, {mode: parity block})."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import segan_oracle as O
    from segan_pytorch_amd import ops
    from segan_pytorch_amd.datasets import synthetic_pairs
    from segan_pytorch_amd.models import WSEGAN
    random.seed(111); np.random.seed(111); torch.manual_seed(111)
    m = WSEGAN(SimpleNamespace(**opts))
    gsd0 = {k: v.detach().clone() for k, v in m.G.state_dict().items()}
    dsd0 = {k: v.detach().clone() for k, v in m.D.state_dict().items()}
    clean, noisy = synthetic_pairs(B, 16384, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(B, 1024, 16, generator=torch.Generator().manual_seed(0))
    rolls = [[1, -2, 3, -4, 5], [-3, 2, -1, 5, 4], [4, 1, -2, 2, -5], [2, -5, 1, -1, -4]]
    names = ['utt_additive_{}'.format(i) if i % 2 == 0 else 'utt_{}'.format(i) for i in range(B)]
    random.seed(77)
    perm = list(range(B))
    random.shuffle(perm)
    st = opts['genc_poolings']
    torch.backends.mkldnn.enabled = False
    O.wsegan_step(gsd0, dsd0, clean[:4], noisy[:4], z[:4], rolls, [1, 0, 3, 2], names[:4], st)  # warm-up
    t0 = time.perf_counter()
    ref = O.wsegan_step(gsd0, dsd0, clean, noisy, z, rolls, perm, names, st, l1_weight=100.0,
                        pow_weight=opts['pow_weight'], lr=5e-5, n_fft=opts['n_fft'])
    dt = time.perf_counter() - t0
    base = dict(value=B / dt, unit='chunks/s', cores=torch.get_num_threads(), nproc=os.cpu_count(),
                kind='port', sample='oracle WSEGAN step (--misalign_pair) at batch {}: warm-up at batch '
                                    '4, ONE timed step, oneDNN off: {:.2f} s'.format(B, dt))
    out = {}
    for name, det in (('fp32_deterministic', True), ('fp32_default', False)):
        old = ops.get_deterministic()
        ops.set_deterministic(det)
        try:
            mm = WSEGAN(SimpleNamespace(**opts))
            mm.G.load_state_dict(gsd0)
            mm.D.load_state_dict(dsd0)
            mm = mm.to(dev)
            Gopt, Dopt = mm.build_optimizers(SimpleNamespace(**opts))
            mm.G.train(); mm.D.train()
            it = iter(rolls)
            mm.D.draw_rolls = lambda: list(next(it))
            cg, ng, zg = clean.to(dev), noisy.to(dev), z.to(dev)
            random.seed(77)                 # the misalign permutation
            Genh, d_loss = mm.wgan_d_phase(cg, ng, Dopt, z=zg)
            y, yr = Genh.detach().cpu().double(), ref['Genh'].double()
            dn = dict(mm.D.named_parameters())
            d_grad = max(_rel(dn[k].grad, g) for k, g in ref['d_grads'].items()
                         if not k.endswith('conv.bias'))
            mm.D.load_state_dict({k: ref['D'].get(k, v) for k, v in dsd0.items()})
            ops.bump_weights_epoch()
            G_cost, g_adv, pow_loss, den_loss = mm.wgan_g_phase(names, Genh, cg, ng, Gopt, 100.0)
            torch.cuda.synchronize()
            gn = dict(mm.G.named_parameters())
            g_grad = max(_rel(gn[k].grad, g) for k, g in ref['g_grads'].items())
            out[name] = {'batch': B, 'g_mse': ((y - yr) ** 2).mean().item(),
                         'g_max_abs': (y - yr).abs().max().item(),
                         'd_loss_rel': _lrel(d_loss, ref['d_loss']), 'g_adv_loss_rel': _lrel(g_adv, ref['g_adv']),
                         'pow_loss_rel': _lrel(pow_loss, ref['pow_loss']),
                         'den_loss_rel': _lrel(den_loss, ref['den_loss']),
                         'd_grad_rel_l2_worst_tensor': d_grad, 'g_grad_rel_l2_worst_tensor': g_grad}
            del mm, Gopt, Dopt
        except Exception as e:
            out[name] = {'error': repr(e)}
        finally:
            ops.set_deterministic(old)
    return base, out


def host_side_costs(dev, zshape, world, backend, device_z):
    """Per-rank host cost of one step's z: all ranks draw at the same moment (after a barrier), 3
    draws each into a pinned buffer, the median; then the H2D copy of that buffer, event-timed.
    Gathered over the ranks.  With --device-z no host draw happens in the step: reported as such."""
    import torch.distributed as dist
    from segan_pytorch_amd import distributed as sdist
    pin = torch.empty(zshape, pin_memory=True)
    dst = torch.empty(zshape, device=dev)
    dist.barrier()
    draws = []
    for _ in range(3):
        t0 = time.perf_counter()
        torch.randn(zshape, out=pin)
        draws.append(1e3 * (time.perf_counter() - t0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(pin, non_blocking=True)       # warm-up
    torch.cuda.synchronize()
    e0.record()
    dst.copy_(pin, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    mine = torch.tensor([sorted(draws)[1], e0.elapsed_time(e1)], dtype=torch.float64,
                        device=dev if backend == 'nccl' else 'cpu')
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return {'z_draw_ms_per_rank': [round(float(t[0]), 2) for t in allr],
            'z_h2d_ms_per_rank': [round(float(t[1]), 3) for t in allr],
            'z_bytes': int(np.prod(zshape)) * 4,
            'z_in_the_timed_steps': 'device generator (no host draw)' if device_z else
                                    'host randn one step ahead on a thread + pinned H2D on a side stream',
            'pinning': sdist.host_pin(), 'torch_threads': torch.get_num_threads(),
            'cpus_visible': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()}


def csrc_sha():
    """Short hash over the HIP sources: profile-derived figures carry the hash of the sources they
    were measured on, so a bench line built on newer kernels shows them as stale."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, 'segan_pytorch_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(ROOT, 'segan_pytorch_amd', 'csrc', '*.h'))):
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:12]


def _profile(name):
    """The newest committed profiles/rNN_<name> file (round-numbered)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_' + name)))
    return c[-1] if c else None


def pmc_mfma_busy(family, suffix=''):
    """MFMA-pipe busy fraction of a kernel family from the newest committed SQ counter passes
    (profiles/rNN_sq_counters.json, scripts/pmc_sq.sh): MFMA instructions x 64 cycles over the
    SIMD-cycles of the launch."""
    try:
        d = json.load(open(_profile('sq_counters{}.json'.format(suffix))))
        return d['_summary'][family + '_mfma_pipe_busy_mean']
    except Exception:
        return None


def pmc_traffic(main=('corr2_kernel', 'corr_kernel<', 'conv_dgrad_short_kernel'),
                extra=('corr_fixup_kernel',), suffix=''):
    """HBM bytes per launch of the dominant kernel family from the newest committed rocprofv3 PMC
    passes (profiles/rNN_pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE in separate passes over
    this same command; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md): bytes
    of the contraction kernels plus their stream-K fix-up passes, per contraction launch.
    Returns (bytes per launch, provenance dict) — the provenance says which file, which source
    hash it was measured on and whether the sources have changed since (`stale`)."""
    path = _profile('pmc_hbm_traffic{}.json'.format(suffix))
    if path is None:
        return None, None
    try:
        d = json.load(open(path))
        ks = d['kernels']
        n = f = w = 0.0
        for name, v in ks.items():
            is_main = any(m in name for m in main)
            if is_main or any(m in name for m in extra):
                n += v['launches'] if is_main else 0
                f += v['launches'] * v['fetch_kb_avg']
                w += v['launches'] * v['write_kb_avg']
        if n == 0:
            return None, None
        prov = {'file': os.path.relpath(path, ROOT), 'measured_on_csrc_sha': d.get('csrc_sha'),
                'current_csrc_sha': csrc_sha()}
        prov['stale'] = prov['measured_on_csrc_sha'] != prov['current_csrc_sha']
        return (2.0 * f + w) * 1024.0 / n, prov
    except Exception:
        return None, None


def make_workload(shape, wsegan, dev, rank, B, device_z=False):
    """Build one benchmark workload on `dev`: the nets (random init, seed 111), their optimizers, a
    resident synthetic batch and `one_step()` = the full GAN step exactly as the training loop runs it
    (z=None: the Generator draws z itself).  Returns a namespace."""
    from segan_pytorch_amd import distributed as sdist
    from segan_pytorch_amd import losses
    from segan_pytorch_amd.datasets import synthetic_pairs
    from segan_pytorch_amd.models import SEGAN, WSEGAN
    opts = default_opts()
    if shape == 'vanilla11':
        opts.update(VANILLA11)
    gflop = gflop_per_chunk(opts, wsegan=wsegan)
    gflop_exec = gflop_per_chunk(opts, wsegan=wsegan, executed=True)
    random.seed(111); np.random.seed(111); torch.manual_seed(111)
    if wsegan:
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
    clean, noisy = synthetic_pairs(B, 16384, seed=rank, device=dev)
    clean, noisy = clean.unsqueeze(1).contiguous(), noisy.unsqueeze(1).contiguous()
    random.seed(1000 + rank)
    torch.manual_seed(2000 + rank)
    if device_z:
        model.G.z_generator = torch.Generator(device=dev).manual_seed(rank)
    else:
        model.G.z_prefetch = z_lookahead_ok(wsegan)     # as the training loops run it: next z one step ahead
    names = ['utt_additive_{}'.format(i) if i % 2 == 0 else 'utt_{}'.format(i) for i in range(B)]
    from segan_pytorch_amd.models.model import _freeze_gc
    _freeze_gc()        # as SEGAN.train / WSEGAN.train do when they start

    def one_step():
        # z=None: Generator.forward draws it, exactly as inside train.py's loop
        if wsegan:
            return model.wgan_step(names, clean, noisy, Gopt, Dopt, 100.0, z=None)
        return model.gan_step(clean, noisy, Gopt, Dopt, criterion, 100.0, z=None)

    return SimpleNamespace(opts=opts, gflop=gflop, gflop_exec=gflop_exec, model=model, Gopt=Gopt, Dopt=Dopt,
                           criterion=criterion, clean=clean, noisy=noisy, names=names, one_step=one_step,
                           shape=shape, wsegan=wsegan)


def measure_host(one_step, reps=5):
    """What the HOST side of a step costs, and what the GPU side costs when the host is never the one
    being waited for (round-5 review, weak 4 / next 3c):

    * host_enqueue_ms_per_step — wall time of one_step() with the device synchronised BEFORE it and not
      after: python + ctypes + the caching allocator + the HIP launches of one step, no GPU in the way
      (median of `reps`; any hidden device sync inside the step would show up here as a step time);
    * gpu_ms_per_step_unstarved — HIP-event time of one step whose launches were ALL enqueued while the
      device was still busy (behind a torch.cuda._sleep blocker sized above the enqueue time of two
      steps, and behind a first, untimed step): the step as the GPU executes it back to back, i.e. what
      a kernel trace's per-step kernel sum + dispatch gaps would read.  `unstarved_ok` says the host
      did finish enqueueing both steps before the blocker ended.

    The timed steps' ms_per_step minus gpu_ms_per_step_unstarved is the time per step the GPU idled
    waiting for the host (`gpu_idle_ms_per_step` in the line)."""
    torch.cuda.synchronize()
    enq = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_step()
        enq.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    enq_med = sorted(enq)[len(enq) // 2]
    # calibrate the blocker: cycles per millisecond of torch.cuda._sleep
    cyc = 20000000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(cyc); e1.record()
    torch.cuda.synchronize()
    per_ms = cyc / max(e0.elapsed_time(e1), 1e-3)
    # TWO steps behind the blocker, the second one timed: it starts the moment the first ends — warm
    # clocks, like every step of the timed loop (a step that follows the idle blocker runs ~1.5 ms
    # slower: the matrix clocks ramp up under the first contraction launches)
    block_ms = 3.0 * max(enq) + 20.0
    gpu, ok = [], True
    for _ in range(3):
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        torch.cuda._sleep(int(block_ms * per_ms))
        one_step()
        s0.record()
        one_step()
        s1.record()
        host_ms = 1e3 * (time.perf_counter() - t0)
        torch.cuda.synchronize()
        gpu.append(s0.elapsed_time(s1))
        ok = ok and host_ms < block_ms
    return {'host_enqueue_ms_per_step': enq_med, 'host_enqueue_ms_all': [round(e, 2) for e in enq],
            'gpu_ms_per_step_unstarved': sorted(gpu)[len(gpu) // 2],
            'gpu_ms_unstarved_all': [round(g, 3) for g in gpu], 'blocker_ms': block_ms, 'unstarved_ok': ok}


def roofline_blocks(summary, peak_tf, ms_per_step, fp32_run=True):
    """`roofline` / `roofline_wgrad` sub-blocks of a workload from the live KernelTimer summary."""
    out = {}
    for fam, key, what in (('corr', 'roofline', 'conv/deconv forward + data gradient'),
                           ('wgrad', 'roofline_wgrad', 'weight gradients')):
        r = summary.get(fam)
        if r:
            out[key] = {'bound': 'mfma', 'kernel': what + (' (fp32 MFMA)' if fp32_run else ' (bf16 MFMA)'),
                        'achieved': r['tflops'], 'peak': peak_tf, 'unit': 'TFLOP/s',
                        'frac': r['tflops'] / peak_tf, 'avg_launch_us': r['avg_us'], 'launches': r['launches'],
                        'launches_per_step': r['launches_per_step'], 'sampled_steps': r['sampled_steps'],
                        'gflop_per_launch': r['flops_per_launch'] / 1e9,
                        'share_of_step_time': r['ms_per_step'] / ms_per_step, 'traffic': None}
    return out


def side_workload(shape, wsegan, dev, rank, world, B, steps, warmup, barrier, device_z=False):
    """One of BASELINE.json's other fp32 configurations timed like the headline (same barrier /
    max-over-ranks protocol, same KernelTimer), for the `other_workloads` block of the line: config 4
    (WSEGAN --misalign_pair) and the literal 11-layer stride-2 shape of config 2."""
    w = make_workload(shape, wsegan, dev, rank, B, device_z)
    try:
        timer = KernelTimer()
        timer.install()
        timer.active = False
        dt, lo = run_timed(w.one_step, steps, warmup, barrier, timer)
        timer.uninstall()
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        ms = 1e3 * dt / steps
        value = B * world * steps / dt
        booked = timer.booked_flops_per_step() / B / 1e9
        out = {'workload': ('WSEGAN step with --misalign_pair (model.py:577-669; BASELINE config 4)' if wsegan
                            else 'original SEGAN shape: 11+11 layers of stride 2, k31 (train.py:199-205 flags; '
                                 'the shape BASELINE config 2 words as "11-layer enc/dec")' if shape == 'vanilla11'
                            else 'SEGAN+ default G+D') + ', batch {} x 16384-sample chunks per GPU, fp32, '
                                                         'full GAN step, RMSprop'.format(B),
               'value': value, 'unit': 'chunks/s', 'ms_per_step': ms, 'steps': steps, 'warmup': warmup,
               'n_gpus': world, 'gflop_per_chunk': w.gflop, 'gflop_per_chunk_executed': booked,
               'step_tflops': w.gflop * value / 1e3,
               'step_frac_of_f32_mfma_peak': w.gflop * value / 1e3 / PEAK_F32_MFMA_TF / world,
               'step_frac_executed': booked * value / 1e3 / PEAK_F32_MFMA_TF / world,
               'losses_finite': all(bool(torch.isfinite(x)) for x in lo),
               'z': 'device generator' if device_z else 'host randn one step ahead on a host thread + H2D'}
        out.update(roofline_blocks(timer.summary(), PEAK_F32_MFMA_TF, ms))
        if 'roofline' in out:
            # HBM bytes per contraction launch from this workload's OWN committed rocprofv3 PMC passes
            tr, prov = pmc_traffic(suffix=('_vanilla11' if shape == 'vanilla11' else '') + ('_wsegan' if wsegan else ''))
            out['roofline']['traffic'], out['roofline']['traffic_source'] = tr, prov
            sfx = ('_vanilla11' if shape == 'vanilla11' else '') + ('_wsegan' if wsegan else '')
            out['roofline']['mfma_pipe_busy_pmc'] = pmc_mfma_busy('corr2', sfx)
            if 'roofline_wgrad' in out:
                out['roofline_wgrad']['mfma_pipe_busy_pmc'] = pmc_mfma_busy('wgrad2', sfx)
        if world == 1:
            h = measure_host(w.one_step)
            out['host'] = h
            out['host_enqueue_ms_per_step'] = h['host_enqueue_ms_per_step']
            out['gpu_ms_per_step_unstarved'] = h['gpu_ms_per_step_unstarved']
            out['gpu_idle_ms_per_step'] = ms - h['gpu_ms_per_step_unstarved']
        return out
    finally:
        from segan_pytorch_amd import distributed as sdist
        w.model.G.cancel_z_prefetch()
        sdist.drop_reducer(w.Gopt)
        sdist.drop_reducer(w.Dopt)
        del w
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=300, help='chunks per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=300, help='batch of the timed CPU oracle steps')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed CPU oracle steps per oneDNN setting (>= 3: SURVEY.md 8d)')
    ap.add_argument('--cpu-budget-s', type=float, default=150.0,
                    help='once the CPU baseline has run this long a oneDNN setting stops at two timed steps')
    ap.add_argument('--device-z', action='store_true',
                    help='draw z on the GPU (train.py --device_z) instead of on the host like the '
                         'reference (generator.py:197); the default times what train.py runs')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--no-modes', action='store_true', help='skip the bf16x3 / bf16 side measurements')
    ap.add_argument('--no-side-workloads', action='store_true',
                    help='skip the `other_workloads` block (WSEGAN = BASELINE config 4, the 11-layer stride-2 '
                         'shape = config 2 as worded) that the default fp32 SEGAN+ run carries')
    ap.add_argument('--side-steps', type=int, default=10, help='timed steps per side workload')
    ap.add_argument('--no-host-measure', action='store_true',
                    help='skip measure_host (host enqueue time per step, GPU time of an unstarved step)')
    ap.add_argument('--comm-ab', action='store_true',
                    help='world > 1: also time the steps with the OTHER gradient transport (libsegan_hip\'s '
                         'own RCCL communicators vs torch.distributed); opt-in — the native transport has '
                         'only ever run at one rank, and a first multi-GPU run should not depend on it')
    ap.add_argument('--wsegan', action='store_true',
                    help='time the WSEGAN step of BASELINE config 4 (--wsegan --misalign_pair) instead '
                         'of the SEGAN+ step; a side measurement, not the headline metric')
    ap.add_argument('--shape', default='segan_plus', choices=['segan_plus', 'vanilla11'],
                    help='segan_plus: the SEGAN+ default net (the headline configuration); vanilla11: '
                         'the original 11-layer stride-2 SEGAN (train.py:199-205 flags) — a side line '
                         'with its own FLOP count')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16x3', 'bf16'],
                    help='forward/data-gradient contraction precision (default: exact fp32, the '
                         'BASELINE configuration; bf16x3 = exact 3-way bf16 split of fp32 operands; '
                         'bf16 = BASELINE config 5)')
    args = ap.parse_args()

    # one rank per GPU: when nobody launched the ranks for us, do it ourselves
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        if args.gpus > torch.cuda.device_count() and 'SEGAN_LOCAL_DEVICE' not in os.environ:
            raise SystemExit('--gpus {} but only {} GPU(s) visible'.format(args.gpus, torch.cuda.device_count()))
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.execvp(cmd[0], cmd)

    from segan_pytorch_amd import distributed as sdist
    from segan_pytorch_amd import losses
    from segan_pytorch_amd.datasets import synthetic_pairs
    from segan_pytorch_amd.models import SEGAN, WSEGAN

    from segan_pytorch_amd import ops as _ops
    _ops.set_precision(args.precision)
    if args.gpus > torch.cuda.device_count() and 'SEGAN_LOCAL_DEVICE' not in os.environ:
        raise SystemExit('--gpus {} but only {} GPU(s) visible'.format(args.gpus, torch.cuda.device_count()))
    rank, world, local = sdist.init_from_env()
    if 'SEGAN_LOCAL_DEVICE' in os.environ:      # test hook: several ranks share one GPU (gloo)
        local = int(os.environ['SEGAN_LOCAL_DEVICE'])
    if world != max(1, args.gpus):
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
    dev = torch.device('cuda', local if world > 1 else 0)
    torch.cuda.set_device(dev)
    ranks_seen, devices_seen, backend = [0], [dev.index], None
    if world > 1:
        import torch.distributed as dist
        backend = dist.get_backend()
        if backend != 'nccl' and 'SEGAN_DIST_BACKEND' not in os.environ:
            raise SystemExit('expected the RCCL (nccl) backend, got {}'.format(backend))
        # (gloo — the shared-GPU test hook — gathers host tensors only)
        mine = torch.tensor([rank, dev.index], device=dev if backend == 'nccl' else 'cpu',
                            dtype=torch.int64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks_seen = sorted(int(t[0]) for t in allr)
        devices_seen = [int(t[1]) for t in allr]
        assert ranks_seen == list(range(world)), ranks_seen

    B = args.batch
    wl = make_workload(args.shape, args.wsegan, dev, rank, B, args.device_z)
    opts, gflop, gflop_exec = wl.opts, wl.gflop, wl.gflop_exec
    model, Gopt, Dopt, one_step = wl.model, wl.Gopt, wl.Dopt, wl.one_step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    timer = None
    if not args.no_kernel_timer:
        timer = KernelTimer()
        timer.install()
        timer.active = False
    # world > 1: two event records per optimizer step of the TIMED steps (comm_stats below)
    dt, losses_out = run_timed(one_step, args.steps, args.warmup, barrier, timer,
                               on_timed_start=(lambda: sdist.set_profile(True)) if world > 1 else None)
    if timer is not None:
        timer.uninstall()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    finite = all(bool(torch.isfinite(x)) for x in losses_out)
    # peak device memory of the warm-up + timed steps (no cyclic collection runs inside the timed region:
    # the step must not depend on the collector to release its activations)
    peak_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    if timer is not None and timer.sampled > 0:
        # what the engine EXECUTED, from the FLOPs the timed launches were booked with (round-5 review,
        # weak 7: the formula's "one D forward less" missed the skipped z half of dec0's data gradient
        # and the first layers' data gradients)
        gflop_exec = timer.booked_flops_per_step() / B / 1e9
    host = None
    if world == 1 and not args.no_host_measure:
        try:
            host = measure_host(one_step)
        except Exception as e:      # a side measurement must never cost the headline line
            host = {'error': repr(e)}

    # ---- the first multi-GPU run diagnoses itself (round-3 review, item 6): what the gradient
    # exchange cost in the timed steps, and the same steps with the OTHER transport ----
    comm = None
    if world > 1:
        comm = sdist.comm_stats()
        sdist.set_profile(False)
        for a in comm['arenas']:
            a['wait_device_ms_per_step'] = a.pop('wait_device_ms') / args.steps
            a['wait_host_ms_per_step'] = a.pop('wait_host_ms') / args.steps
        # max over ranks of the device time the compute stream waited for collectives per step
        w = torch.tensor([sum(a['wait_device_ms_per_step'] for a in comm['arenas'])],
                         device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(w, op=torch.distributed.ReduceOp.MAX)
        comm['comm_wait_ms_per_step'] = float(w.item())
        comm['ms_per_step'] = {('native' if sdist.native_comm() is not None else 'torch.distributed'):
                               1e3 * dt / args.steps}
        # what every rank's HOST side costs with all ranks of the node at it at once (round-4 review,
        # item 7): the single-threaded randn of one z (drawn one step ahead on a host thread: it must
        # stay below the step time) and its pinned H2D copy, per rank; plus how the ranks were pinned
        comm['host'] = host_side_costs(dev, (B, opts['z_dim'], 16384 // int(np.prod(opts['genc_poolings']))),
                                       world, backend, args.device_z)
        if backend == 'nccl' and args.comm_ab:
            was_native = sdist.native_comm() is not None
            try:
                sdist.set_native(not was_native)
                for _ in range(2):
                    one_step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                barrier()
                dd = time.perf_counter() - t1
                t = torch.tensor([dd], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                comm['ms_per_step']['torch.distributed' if was_native else 'native'] = \
                    1e3 * float(t.item()) / args.steps
            except Exception as e:      # a side measurement must never cost the headline line
                comm['other_transport_error'] = repr(e)
            finally:
                sdist.set_native(was_native)

    # what reproducibility costs: the same K steps in the OTHER reduction mode (the default —
    # and the timed one — adds the weight-gradient / dense-head contraction splits with fp32
    # atomics; the deterministic mode adds them in a fixed order)
    ms_other = None
    timed_det = _ops.get_deterministic()
    if not args.no_modes:
        _ops.set_deterministic(not timed_det)
        try:
            for _ in range(2):
                one_step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            barrier()
            dd = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dd], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                dd = float(t.item())
            ms_other = 1e3 * dd / args.steps
        finally:
            _ops.set_deterministic(timed_det)

    # the fp32 step with blocked accumulation (ops.set_accumulation: 5x lower forward error for
    # one resident wave per SIMD), timed beside the headline
    ms_blocked = None
    if args.precision == 'fp32' and not args.no_modes:
        _ops.set_accumulation('blocked')
        try:
            for _ in range(2):
                one_step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            barrier()
            dd = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dd], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                dd = float(t.item())
            ms_blocked = 1e3 * dd / args.steps
        finally:
            _ops.set_accumulation('plain')

    # Same step with the contractions on the bf16 matrix cores, reported BESIDE the fp32
    # headline (never as `value`): 'bf16x3' = exact 3-way split of the fp32 operands,
    # 'bf16' = BASELINE config 5.  Every rank runs the same steps (the collectives match).
    modes = {}
    if args.precision == 'fp32' and not args.no_modes:
        for prec in ('bf16x3', 'bf16'):
            host_gen = None
            try:
                _ops.set_precision(prec)
                if prec == 'bf16' and not args.device_z:
                    # as train.py --precision bf16 runs it: z drawn on the GPU — a bf16 step takes about
                    # as long as the single-threaded host randn of one z (~25 ms), which would
                    # otherwise be the critical path (train.py --host_z restores the host draw)
                    model.G.cancel_z_prefetch()
                    model.G.z_prefetch = False
                    host_gen = True
                    model.G.z_generator = torch.Generator(device=dev).manual_seed(rank)
                mt = None
                if not args.no_kernel_timer:
                    mt = KernelTimer()
                    mt.install()
                    mt.active = False
                dm, lo = run_timed(one_step, args.steps, 2, barrier, mt)
                if mt is not None:
                    mt.uninstall()
                if world > 1:
                    t = torch.tensor([dm], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    dm = float(t.item())
                modes[prec] = {'value': B * world * args.steps / dm, 'unit': 'chunks/s',
                               'ms_per_step': 1e3 * dm / args.steps,
                               'z': 'device generator (train.py --precision bf16 default)'
                                    if (host_gen or args.device_z) else 'host randn one step ahead + H2D',
                               'losses_finite': all(bool(torch.isfinite(x)) for x in lo)}
                if mt is not None:
                    # bf16: one bf16 MFMA per product; bf16x3: six, so the peak
                    # in fp32-equivalent FLOPs is a sixth of the dense bf16 peak
                    peak = PEAK_BF16_MFMA_TF / (6.0 if prec == 'bf16x3' else 1.0)
                    for fam, key in (('corr', 'roofline'), ('wgrad', 'roofline_wgrad')):
                        r = mt.summary().get(fam)
                        if r:
                            modes[prec][key] = {
                                'bound': 'mfma', 'achieved': r['tflops'], 'peak': peak, 'unit': 'TFLOP/s',
                                'frac': r['tflops'] / peak, 'avg_launch_us': r['avg_us'],
                                'launches': r['launches'], 'launches_per_step': r['launches_per_step'],
                                'share_of_step_time': r['ms_per_step'] / (1e3 * dm / args.steps),
                                'kernel': ('conv/deconv forward + data gradient' if fam == 'corr'
                                           else 'weight gradients') + ' on v_mfma_f32_32x32x16_bf16'}
            except Exception as e:      # a side measurement must never cost the headline line
                modes[prec] = {'error': repr(e)}
                break
            finally:
                _ops.set_precision('fp32')
                if host_gen:
                    model.G.z_generator = None
                    model.G.z_prefetch = z_lookahead_ok(args.wsegan)

    # BASELINE.json's other fp32 configurations, timed like the headline and carried by the SAME line
    # (round-5 review, next 1): config 4 (WSEGAN --misalign_pair) and config 2's literal 11-layer shape
    side = {}
    if args.precision == 'fp32' and args.shape == 'segan_plus' and not args.wsegan and \
            not args.no_side_workloads:
        model.G.cancel_z_prefetch()
        for key, shp, ws_ in (('wsegan', 'segan_plus', True), ('vanilla11', 'vanilla11', False)):
            try:
                side[key] = side_workload(shp, ws_, dev, rank, world, B, args.side_steps, args.warmup,
                                          barrier, args.device_z)
            except Exception as e:      # a side measurement must never cost the headline line
                side[key] = {'error': repr(e)}

    if rank == 0:
        chunks = B * world * args.steps
        value = chunks / dt
        ms = 1e3 * dt / args.steps
        # the matrix-core peak of the mode that was timed: fp32 MFMA, the dense bf16 MFMA, or — bf16x3,
        # six bf16 MFMAs per product — a sixth of it in fp32-equivalent FLOPs
        fp32_run = args.precision == 'fp32'
        peak_tf = (PEAK_F32_MFMA_TF if fp32_run else
                   PEAK_BF16_MFMA_TF / (6.0 if args.precision == 'bf16x3' else 1.0))
        line = {
            'metric': '16384-sample waveform chunks/sec (GAN step)', 'value': value,
            'unit': 'chunks/s', 'n_gpus': world, 'ranks_seen': ranks_seen, 'devices': devices_seen,
            'backend': backend, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16x3': 'f32 operands as 3 bf16 planes (6 products), f32 accumulate',
                      'bf16': 'bf16 operands, f32 accumulate'}[args.precision], 'data': 'synthetic',
            'config': {'workload': ('WSEGAN step with --misalign_pair (model.py:577-669; BASELINE '
                                    'config 4), same nets, batch {} x 16384-sample chunks per GPU'
                                    if args.wsegan else
                                    'original SEGAN shape: 11+11 layers of stride 2, k31 (train.py:'
                                    '199-205 flags), batch {} x 16384-sample chunks per GPU, full GAN '
                                    'step, RMSprop, {} (side line, not the headline configuration)'
                                    if args.shape == 'vanilla11' else
                                    'SEGAN+ default G+D (5+5 layers, k31, stride 4, z 1024x16), '
                                    'batch {} x 16384-sample chunks per GPU, full GAN step '
                                    '(model.py:292-321), RMSprop, {}').format(
                                        B, 'fp32' if fp32_run else args.precision +
                                        ' contractions (side configuration, not the headline)'),
                       'global_batch': B * world, 'parallelism': 'dp{}'.format(world),
                       'z': 'device generator' if args.device_z else 'host randn (one step ahead on a host thread, as SEGAN.train) + H2D per step'},
            'losses_finite': finite,
            'max_memory_allocated_gb': peak_gb,
            'comm': comm,
            'comm_wait_ms_per_step': comm['comm_wait_ms_per_step'] if comm else None,
            'precision': args.precision,
            'reduction_mode': ('deterministic (fixed-order reductions, SEGAN_DETERMINISTIC=1)' if timed_det else
                               'default: fp32 atomics in the weight-gradient / dense-head contraction splits'),
            'ms_per_step_deterministic': ms if timed_det else ms_other,
            'ms_per_step_atomics': ms_other if timed_det else ms,
            'ms_per_step_blocked_accumulation': ms_blocked,
            'accumulation': _ops.get_accumulation(),
            'gflop_per_chunk': gflop,
            'gflop_per_chunk_executed': gflop_exec,
            'step_tflops': gflop * value / 1e3,
            'step_peak_tflops': peak_tf,
            'step_frac_of_f32_mfma_peak': (gflop * value / 1e3 / PEAK_F32_MFMA_TF / world
                                           if args.precision == 'fp32' else None),
            'step_frac_of_mfma_peak': gflop * value / 1e3 / peak_tf / world,
            'step_frac_executed': gflop_exec * value / 1e3 / peak_tf / world,
            'step_frac_note': 'step_frac_of_f32_mfma_peak divides the REFERENCE accounting (SURVEY.md 8d: '
                              'it counts the D weight gradients of the generator phase, which the reference '
                              'computes and discards) by the time; step_frac_executed counts only what this '
                              'engine executes: the FLOPs its timed launches were booked with (no D weight '
                              'gradients in the generator phase, no z half of dec0\'s data gradient, no '
                              'first-layer data gradients)',
            'gflop_per_chunk_executed_source': ('FLOPs booked by the timed launches (conv/deconv forward, data and '
                                                'weight gradients, dense-head / STFT GEMMs)'
                                                if timer is not None else 'formula (no kernel timer)'),
            'host': host,
            'host_enqueue_ms_per_step': host.get('host_enqueue_ms_per_step') if host else None,
            'gpu_ms_per_step_unstarved': host.get('gpu_ms_per_step_unstarved') if host else None,
            'gpu_idle_ms_per_step': (ms - host['gpu_ms_per_step_unstarved']
                                     if host and 'gpu_ms_per_step_unstarved' in host else None),
            'step_hbm_gbs_algorithmic': MB_PER_CHUNK * value / 1e3 / world,
            'step_frac_of_hbm_roofline': MB_PER_CHUNK * value / 1e3 / world / PEAK_HBM_GBS,
        }
        if timer is not None:
            s = timer.summary()
            c = s.get('corr')
            # counters belong to the workload they were collected on (round-4 review, weak 8: the
            # 11-layer side line used to carry the SEGAN+ profile's figures): the committed PMC
            # files are looked up by workload suffix and a workload without its own passes gets null
            wsuf = ('_vanilla11' if args.shape == 'vanilla11' else '') + ('_wsegan' if args.wsegan else '')
            if fp32_run:
                traffic, traffic_prov = pmc_traffic(suffix=wsuf)
            else:       # the bf16 profile exists for 'bf16' only
                traffic, traffic_prov = (pmc_traffic(main=('corr_bf2_kernel', 'corr_bf_kernel'),
                                                     extra=('bf2_fixup_kernel', 'act_pack_kernel'),
                                                     suffix='_bf16' + wsuf)
                                         if args.precision == 'bf16' else (None, None))
            if c:
                line['roofline'] = {
                    'bound': 'mfma',
                    'kernel': ('corr2_kernel + conv_dgrad_short_kernel (conv/deconv forward + data gradient)'
                               if fp32_run else 'conv/deconv forward + data gradient entry points on '
                               'v_mfma_f32_32x32x16_bf16 (corr_bf2_kernel + its packing / fix-up passes)'),
                    'achieved': c['tflops'], 'peak': peak_tf, 'unit': 'TFLOP/s',
                    'frac': c['tflops'] / peak_tf, 'traffic': traffic,
                    'traffic_unit': 'HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; a committed '
                                    'profile, not measured in this run: see traffic_source)',
                    'traffic_source': traffic_prov,
                    'avg_launch_us': c['avg_us'], 'launches': c['launches'],
                    'gflop_per_launch': c['flops_per_launch'] / 1e9,
                    'launches_per_step': c['launches_per_step'], 'sampled_steps': c['sampled_steps'],
                    'share_of_step_time': c['ms_per_step'] / ms,
                    'mfma_pipe_busy_pmc': (pmc_mfma_busy('corr2', wsuf) if fp32_run else
                                           pmc_mfma_busy('corr_bf2', '_bf16' + wsuf) if args.precision == 'bf16'
                                           else None)}
            if 'wgrad' in s:
                w = s['wgrad']
                line['roofline_wgrad'] = {'bound': 'mfma',
                                          'kernel': 'wgrad2_kernel' if fp32_run else
                                                    'weight-gradient entry point on v_mfma_f32_32x32x16_bf16 '
                                                    '(wgrad_bf2_kernel + its packing passes)',
                                          'achieved': w['tflops'], 'peak': peak_tf,
                                          'unit': 'TFLOP/s', 'frac': w['tflops'] / peak_tf,
                                          'avg_launch_us': w['avg_us'], 'launches': w['launches'],
                                          'launches_per_step': w['launches_per_step'],
                                          'share_of_step_time': w['ms_per_step'] / ms,
                                          'mfma_pipe_busy_pmc': (
                                              pmc_mfma_busy('wgrad2', wsuf) if fp32_run else
                                              pmc_mfma_busy('wgrad_bf2', '_bf16' + wsuf) if args.precision == 'bf16'
                                              else None)}
        if side:
            side['note'] = ('BASELINE.json configs beside the headline, each a full fp32 GAN step at the same '
                            'per-GPU batch, timed with the same barrier / max-over-ranks protocol: `wsegan` = '
                            'config 4 (run_wsegan_train.sh flags), `vanilla11` = the 11-layer stride-2 shape '
                            'config 2 words; roofline blocks from the live HIP-event kernel timer; host_* / '
                            'gpu_* from measure_host (N = 1 only)')
            line['other_workloads'] = side
        if modes:
            modes['note'] = ('same workload, contractions on the bf16 MFMA: bf16x3 = fp32 operands '
                             'split exactly into 3 bf16 planes, 6 partial products, fp32 accumulate '
                             '(parity tolerance 5e-5, tests/test_gpu_kernels.py); bf16 = BASELINE '
                             'config 5 (tolerance 2e-2).  `value` above is the exact-fp32 run.')
            line['other_precisions'] = modes
        if world == 1 and not args.no_cpu_baseline and args.shape == 'segan_plus' and not args.wsegan:
            try:
                del model, Gopt, Dopt, one_step, wl
                torch.cuda.empty_cache()
                line['cpu_baseline'], parity = cpu_baseline(
                    args.cpu_batch, args.cpu_steps, dev, budget_s=args.cpu_budget_s,
                    modes=tuple(k for k in modes if k in ('bf16x3', 'bf16') and 'error' not in modes[k]))
                line['speedup_vs_cpu_baseline'] = value / line['cpu_baseline']['value']
                if parity:
                    line['parity'] = dict(parity.get('fp32_deterministic', {}), note=PARITY_NOTE)
                    line['parity_default_mode'] = parity.get('fp32_default')
                    line['parity_blocked_accumulation'] = parity.get('fp32_blocked')
                    for k in ('bf16x3', 'bf16'):
                        if k in parity and k in line.get('other_precisions', {}):
                            line['other_precisions'][k]['parity'] = parity[k]
            except Exception as e:  # the bench line must still be printed
                line['cpu_baseline'] = {'error': repr(e)}
        elif world == 1 and not args.no_cpu_baseline and args.shape == 'segan_plus' and args.wsegan:
            try:
                del model, Gopt, Dopt, one_step, wl
                torch.cuda.empty_cache()
                line['cpu_baseline'], par = wsegan_parity(opts, args.cpu_batch, dev)
                line['speedup_vs_cpu_baseline'] = value / line['cpu_baseline']['value']
                line['parity'] = dict(par.get('fp32_deterministic', {}),
                                      note='HIP WSEGAN step vs the oracle WSEGAN step timed above (same '
                                           'weights / inputs / z / phase shifts / misalign permutation; '
                                           'generator phase through the oracle\'s post-step D)')
                line['parity_default_mode'] = par.get('fp32_default')
            except Exception as e:
                line['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(line))
    sdist.destroy_native()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
