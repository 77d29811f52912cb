# First GPU call of a round (≈ 2 min of box time): the numbers every later decision starts from.
#   bash scripts/next_round_first.sh r06
# -> gpurun_out/<round>_first/{bench.json, bf16_tile_probe.json, layers_bf16.txt}
set -u
R=${1:-r06}
cd $GRAFT_REPO_ROOT; O=gpurun_out/${R}_first; mkdir -p $O
# headline + deterministic + blocked + bf16x3 + bf16 legs, no CPU oracle (that is 3 of a default run's 4 minutes)
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
# what the LDS-fed bf16 MFMA loop sustains per wave-tile shape (first row)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/bf16_tile_probe scripts/bf16_tile_probe.hip 2> $O/probe_build.err \
  && timeout 60 scripts/bf16_tile_probe > $O/bf16_tile_probe.json 2> $O/probe.err
# every bf16 contraction in isolation
SEGAN_PRECISION=bf16 timeout 120 python scripts/bench_layers.py --iters 3 > $O/layers_bf16.txt 2>&1
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('fp32 %.2f ms  det %.2f  blocked %.2f  corr %.3f  wgrad %.3f' % (
    d['ms_per_step'], d['ms_per_step_deterministic'], d['ms_per_step_blocked_accumulation'],
    d['roofline']['frac'], d['roofline_wgrad']['frac']))
for k, v in d['other_precisions'].items():
    if isinstance(v, dict):
        print(k, '%.2f ms' % v['ms_per_step'], v.get('roofline', {}).get('frac'), v.get('roofline_wgrad', {}).get('frac'))
try:
    for r in json.load(open('$O/bf16_tile_probe.json'))['rows']:
        print(r)
except Exception as e:
    print('probe:', e)
PY
