# Round 6: what the driver runs at round end, in its order — the GPU suite (timed), smoke(), the default bench.
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 ) > $O/tests_full.log 2>&1
echo "tests rc=$?" >> $O/tests_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -c 1800 $O/tests_full.log; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06v/bench_steps20.json').read().splitlines()[0])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'max_memory_allocated_gb', 'gpu_idle_ms_per_step', 'host_enqueue_ms_per_step', 'gflop_per_chunk_executed', 'step_frac_executed')})
print({k: (v.get('value'), v.get('ms_per_step')) for k, v in d['other_workloads'].items() if k != 'note'})
print({k: (v.get('value'), v.get('ms_per_step')) for k, v in d['other_precisions'].items() if k != 'note'})
PY
