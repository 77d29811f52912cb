"""The committed bench line (the newest profiles/rNN_bench_line.json, the stdout of `python
bench.py` on an MI355X) carries every field of the driver's contract."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_bench_line.json')))
    return json.load(open(files[-1])), os.path.basename(files[-1])


def test_committed_bench_line_has_the_contract_fields():
    d, name = _line()
    for k, typ in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int),
                   ('warmup', int), ('ms_per_step', float), ('higher_is_better', bool),
                   ('scaling', str), ('dtype', str), ('data', str), ('config', dict)):
        assert isinstance(d[k], typ), k
    assert 'vs_baseline' in d and d['vs_baseline'] is None       # BASELINE.md publishes no number
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['dtype'] == 'f32'
    assert d['n_gpus'] == 1 and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 300 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['traffic'] > 0
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    assert 'batch 300' in c['sample'] and c['cores'] >= 1
    assert set(d['other_precisions']) >= {'bf16x3', 'bf16'}
    # BASELINE's second metric, at the benchmarked batch: generator output MSE vs the oracle
    pr = d['parity']
    assert pr['batch'] == 300 and pr['g_mse'] < 1e-4 and pr['g_max_abs'] < 1e-5
    assert max(pr['d_real_loss_rel'], pr['d_fake_loss_rel'], pr['g_adv_loss_rel'], pr['g_l1_loss_rel']) < 1e-4
    if name >= 'r03':
        # from round 3 on: the timed (atomics) mode, the deterministic mode and the bf16 modes are all
        # compared with the oracle at the benchmarked batch, and both reduction modes are timed
        pd = d['parity_default_mode']
        assert pd['batch'] == 300 and pd['g_mse'] < 1e-4 and pd['g_max_abs'] < 1e-5
        assert d['ms_per_step_atomics'] == d['ms_per_step'] and d['ms_per_step_deterministic'] > 0
        for k, mse_tol in (('bf16x3', 1e-9), ('bf16', 1e-4)):
            pp = d['other_precisions'][k]['parity']
            assert pp['batch'] == 300 and pp['g_mse'] < mse_tol, k
            assert d['other_precisions'][k]['roofline']['peak'] > 400
        assert d['ms_per_step_deterministic'] > 0
        assert d['gflop_per_chunk_executed'] < d['gflop_per_chunk']
        assert abs(d['step_frac_executed'] - d['gflop_per_chunk_executed'] * d['value'] / 1e3 / 157.3) < 1e-9
        assert 'stale' in r['traffic_source']


def test_flop_accounting():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    o = bench.default_opts()
    assert abs(bench.gflop_per_chunk(o) - 37.96) < 0.01
    assert abs(bench.gflop_per_chunk(o, executed=True) - 35.84) < 0.01
    assert abs(bench.gflop_per_chunk(o, wsegan=True) - 44.33) < 0.01


@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'bf16x3'])
@pytest.mark.parametrize('wsegan,shape', [(False, 'segan_plus'), (True, 'segan_plus'), (False, 'vanilla11')])
def test_bench_line_assembly_runs_without_a_gpu(prec, wsegan, shape):
    """bench.py's rank-0 line assembly (from the timed numbers to the `roofline` blocks) executed on
    fakes: every precision prices its kernels against the peak of the matrix-core mode that ran
    (fp32 MFMA, dense bf16 MFMA, a sixth of it for bf16x3), so no frac can exceed 1 because of the
    wrong denominator, and the line stays JSON-serialisable."""
    import textwrap
    import types
    sys.path.insert(0, ROOT)
    import bench
    src = open(os.path.join(ROOT, 'bench.py')).read()
    seg = textwrap.dedent(src[src.index("        chunks = B * world * args.steps\n"):
                              src.index("        if side:\n            side['note']")])

    class Timer(object):
        def summary(self):
            return {'corr': dict(tflops=100.0, avg_us=100.0, launches=10, flops_per_launch=1e12, total_ms=10.0,
                                 ms_per_step=3.3, launches_per_step=3.3, sampled_steps=3),
                    'wgrad': dict(tflops=90.0, avg_us=100.0, launches=5, total_ms=5.0, ms_per_step=1.7,
                                  launches_per_step=1.7, sampled_steps=3)}

    ns = dict(vars(bench))
    ns.update(B=300, world=1, dt=0.9, ranks_seen=[0], devices_seen=[0], backend=None, finite=True, comm=None,
              timed_det=False, host=None, peak_gb=30.0, ms_other=90.0, ms_blocked=91.0, gflop=37.96, gflop_exec=35.84, timer=Timer(),
              _ops=types.SimpleNamespace(get_accumulation=lambda: 'plain'),
              args=types.SimpleNamespace(steps=10, warmup=3, precision=prec, wsegan=wsegan, shape=shape,
                                         device_z=False))
    exec(seg, ns)
    line = ns['line']
    json.dumps(line)
    peak = {'fp32': 157.3, 'bf16': 2500.0, 'bf16x3': 2500.0 / 6}[prec]
    for key in ('roofline', 'roofline_wgrad'):
        assert abs(line[key]['peak'] - peak) < 1e-6 * peak
        assert abs(line[key]['frac'] - line[key]['achieved'] / peak) < 1e-12
    assert (line['step_frac_of_f32_mfma_peak'] is None) == (prec != 'fp32')
    assert abs(line['step_frac_of_mfma_peak'] - 37.96 * line['value'] / 1e3 / peak) < 1e-12
    assert prec in line['config']['workload'] or wsegan


def test_kernel_timer_samples_three_steps_and_creates_its_events_before_the_timed_region(monkeypatch):
    """bench.KernelTimer / run_timed (round 6: two freshly created timing events around every contraction
    call made the 11-layer shape's step 78.6 instead of 68.1 ms in a fresh process).  On fakes: the last
    warm-up step runs instrumented and creates the pool (three steps' worth of events), exactly the
    first / middle / last timed steps are bracketed — from the pool, no event is created inside the timed
    region —, the per-step figures divide by the sampled steps, and finish() drops the events."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    import torch
    from segan_pytorch_amd import ops
    created = []

    class Ev(object):
        def __init__(self, enable_timing=False):
            created.append(self)
            self.n = 0

        def record(self):
            self.n += 1

        def elapsed_time(self, other):
            return 2.0

    monkeypatch.setattr(torch.cuda, 'Event', Ev)
    calls = []
    monkeypatch.setattr(ops, 'gemm', lambda *a: calls.append(len(created)))
    C = types.SimpleNamespace()
    state = {'in_timed': False, 'created_in_timed': 0}

    def one_step():
        before = len(created)
        ops.gemm(C, 0, 0, C, 0, 0, C, 4, 4, 4, True)
        ops.gemm(C, 0, 0, C, 0, 0, C, 4, 4, 4, True)
        if state['in_timed']:
            state['created_in_timed'] += len(created) - before

    def barrier():
        state['in_timed'] = not state['in_timed']      # first call opens the timed region, second closes it

    t = bench.KernelTimer()
    t.install()
    t.active = False
    dt, _ = bench.run_timed(one_step, 10, 3, barrier, t)
    t.uninstall()
    assert bench.sample_steps(10) == [0, 5, 9] and bench.sample_steps(1) == [0] and bench.sample_steps(2) == [0, 1]
    assert len(calls) == 2 * 13                          # 3 warm-up + 10 timed steps, instrumented or not
    assert len(created) == 12 and state['created_in_timed'] == 0      # 2 calls x 2 events x 3 sampled steps
    s = t.summary()['gemm']
    assert t.sampled == 3 and s['launches'] == 6 and s['launches_per_step'] == 2.0 and s['sampled_steps'] == 3
    assert abs(s['ms_per_step'] - 4.0) < 1e-12 and abs(t.booked_flops_per_step() - 2 * 128.0) < 1e-9
    assert t.records == [] and t._pool == []             # finish(): nothing left for the collector to find
