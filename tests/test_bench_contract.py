"""The committed bench line (profiles/r02_bench_line.json, the stdout of `python bench.py` on
an MI355X) carries every field of the driver's contract."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r02_bench_line.json')))
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
