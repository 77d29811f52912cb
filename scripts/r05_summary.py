"""Key figures of the round-5 evidence files under profiles/ (what DESIGN.md section 5 quotes).
usage: python scripts/r05_summary.py [round-prefix, default r05]"""
import csv
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else 'r05'
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')


def jl(name):
    txt = open(os.path.join(P, name)).read().strip().splitlines()
    return json.loads(txt[-1])


def stats(name, pick):
    tot = {}
    alls = 0.0
    for r in csv.DictReader(open(os.path.join(P, name))):
        ms = float(r['MsPerStep'])
        alls += ms
        for key, subs in pick.items():
            if any(s in r['Name'] for s in subs):
                tot[key] = tot.get(key, 0.0) + ms
                break
    tot['all kernels'] = alls
    return {k: round(v, 3) for k, v in tot.items()}


d = jl(R + '_bench_line.json')
print('headline: %.1f chunks/s, %.2f ms/step; det %.2f, blocked %.2f' % (
    d['value'], d['ms_per_step'], d['ms_per_step_deterministic'], d['ms_per_step_blocked_accumulation']))
print('step_frac_of_f32_mfma_peak %.3f  executed %.3f' % (d['step_frac_of_f32_mfma_peak'], d['step_frac_executed']))
for k in ('roofline', 'roofline_wgrad'):
    r = d[k]
    print(k, 'achieved %.1f TF/s frac %.3f avg %.0f us x %d, share %.3f, traffic %s, busy %s' % (
        r['achieved'], r['frac'], r['avg_launch_us'], r['launches'], r['share_of_step_time'],
        r.get('traffic'), r.get('mfma_pipe_busy_pmc')))
print('traffic_source', d['roofline'].get('traffic_source'))
c = d['cpu_baseline']
print('cpu_baseline', {k: c[k] for k in c if k in ('value', 'cores', 'kind', 'seconds_per_step', 'value_best_step')},
      'speedup %.0f' % d.get('speedup_vs_cpu_baseline', 0))
print('parity', {k: v for k, v in d['parity'].items() if k != 'note'})
print('parity_default_mode', d.get('parity_default_mode'))
for k, v in d['other_precisions'].items():
    if isinstance(v, dict):
        print(k, '%.2f ms %.0f chunks/s' % (v['ms_per_step'], v['value']), v.get('z'),
              'corr %.3f wgrad %.3f' % (v['roofline']['frac'], v['roofline_wgrad']['frac']), v.get('parity'))
pick = {'corr2': ['corr2_kernel'], 'short': ['conv_dgrad_short'], 'wgrad2': ['wgrad2_kernel'],
        'fixup': ['corr_fixup', 'bf2_fixup'], 'act_bwd': ['act_bwd'], 'edge': ['fsmall', 'tsmall', 'wgrad_kernel<'],
        'bn': ['bn_partial', 'bn_final'], 'pack weights': ['pack_f_kernel', 'pack_t_kernel', 'pack_g_kernel', 'pack_bf_'],
        'optim': ['rmsprop', 'adam'], 'head': ['gemm', 'bias_prelu'], 'corr_bf2': ['corr_bf2'], 'wgrad_bf2': ['wgrad_bf2'],
        'act pack': ['act_pack', 'wgrad_pack']}
for n in ('_bench_kernel_stats.csv', '_bench_kernel_stats_bf16.csv', '_bench_kernel_stats_bf16x3.csv',
          '_bench_kernel_stats_vanilla11.csv'):
    if os.path.exists(os.path.join(P, R + n)):
        print(n, stats(R + n, pick))
for n in ('_pmc_hbm_traffic.json', '_pmc_hbm_traffic_bf16.json', '_pmc_hbm_traffic_vanilla11.json'):
    if os.path.exists(os.path.join(P, R + n)):
        t = json.load(open(os.path.join(P, R + n)))
        print(n, 'fetch raw %.2f GB write %.2f GB per step, sha %s' % (
            t['per_step_fetch_gb_raw'], t['per_step_write_gb'], t.get('csrc_sha')))
for n in ('_sq_counters.json', '_sq_counters_bf16.json'):
    if os.path.exists(os.path.join(P, R + n)):
        print(n, json.load(open(os.path.join(P, R + n))).get('_summary'))
for n in ('_bench_line_wsegan.json', '_bench_line_vanilla11.json'):
    if os.path.exists(os.path.join(P, R + n)):
        w = jl(R + n)
        print(n, '%.2f ms %.0f chunks/s corr %.3f wgrad %.3f traffic %s busy %s' % (
            w['ms_per_step'], w['value'], w['roofline']['frac'], w['roofline_wgrad']['frac'],
            w['roofline'].get('traffic'), w['roofline'].get('mfma_pipe_busy_pmc')), w.get('parity'),
            w.get('cpu_baseline', {}).get('value'))
if os.path.exists(os.path.join(P, R + '_train_loop.json')):
    print('train loop', open(os.path.join(P, R + '_train_loop.json')).read()[:600])
