# Round 5, review item 8 (traffic ratio: cut it or prove it free): the short-row conv data gradient
# (conv_dgrad_short_kernel: 9.8 of the step's 42.8 GB raw FETCH_SIZE, 1.19 GB per launch against 84 MB
# of operands) with its tiles in 2-D blocks inside each XCD's range (SEGAN_SHORT_ORDER=1, the
# default) against row-major order (=0): FETCH_SIZE per launch, kernel time, step time.
#   bash scripts/r05_short_order_ab.sh  ->  gpurun_out/r05_short_ab/{summary.json, ...}
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_short_ab; rm -rf $O; mkdir -p $O
for v in 1 0 1 0; do
  SEGAN_SHORT_ORDER=$v python bench.py --no-modes --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'order': $v, 'ms_per_step': d['ms_per_step'], 'corr_frac': d['roofline']['frac']}))" >> $O/steps.jsonl
done
for v in 1 0; do
  SEGAN_SHORT_ORDER=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_$v -o run --output-format csv -- \
    python bench.py --no-modes --no-cpu-baseline --no-kernel-timer --steps 2 --warmup 1 > $O/pmc_$v.log 2>&1
  SEGAN_SHORT_ORDER=$v rocprofv3 --kernel-trace --stats -d $O/kt_$v -o run --output-format csv -- \
    python bench.py --no-modes --no-cpu-baseline --no-kernel-timer --steps 4 --warmup 1 > $O/kt_$v.log 2>&1
done
python - <<PY
import csv, glob, json
out = {'steps': [json.loads(l) for l in open('$O/steps.jsonl')]}
for v in (1, 0):
    f = [float(r['Counter_Value']) for p in glob.glob('$O/pmc_%d/**/*counter_collection.csv' % v, recursive=True)
         for r in csv.DictReader(open(p)) if r['Counter_Name'] == 'FETCH_SIZE' and 'conv_dgrad_short' in r['Kernel_Name']]
    allf = [float(r['Counter_Value']) for p in glob.glob('$O/pmc_%d/**/*counter_collection.csv' % v, recursive=True)
            for r in csv.DictReader(open(p)) if r['Counter_Name'] == 'FETCH_SIZE']
    t = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
         for p in glob.glob('$O/kt_%d/**/*kernel_trace.csv' % v, recursive=True)
         for r in csv.DictReader(open(p)) if 'conv_dgrad_short' in r['Kernel_Name']]
    out['order_%d' % v] = {'short_fetch_mb_per_launch_raw': sum(f) / max(len(f), 1) / 1e3 * 1.024, 'short_launches': len(f),
                           'step_fetch_gb_raw': sum(allf) / 1e6 * 1.024 / 3,
                           'short_kernel_us_avg': sum(t) / max(len(t), 1), 'short_kernel_launches_timed': len(t)}
json.dump(out, open('$O/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_1 $O/pmc_0 $O/kt_1 $O/kt_0
