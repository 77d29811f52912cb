# round 4: per-kernel time of the bf16 step (rocprofv3 kernel trace) + the quick bench lines
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R}q; rm -rf $O; mkdir -p $O
B="python bench.py --no-modes --no-cpu-baseline --no-kernel-timer"
for p in bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$p -o run -- $B --steps 6 --warmup 1 --precision $p > $O/bench_prof_$p.log 2>&1
  python scripts/rocpd_stats.py $O/prof_$p/*results.db $O/kernel_stats_$p.csv 7 > /dev/null 2>&1; rm -rf $O/prof_$p
done
python bench.py --no-cpu-baseline --steps 10 > $O/bench.json 2> $O/bench.err
python - <<PY
import json, csv
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('fp32 %.2f ms det %.2f corr %.3f wgrad %.3f' % (d['ms_per_step'], d['ms_per_step_deterministic'], d['roofline']['frac'], d['roofline_wgrad']['frac']))
for k, v in d['other_precisions'].items():
    if isinstance(v, dict) and 'ms_per_step' in v:
        print(k, '%.2f ms' % v['ms_per_step'], v.get('roofline', {}).get('frac'), v.get('roofline_wgrad', {}).get('frac'))
rows = list(csv.reader(open('$O/kernel_stats_bf16.csv')))
for r in rows[1:26]:
    print('%-80s %5s %9.1f us %7.3f ms/step' % (r[0][:80], r[1], float(r[3]) / 1e3, float(r[7])))
PY
