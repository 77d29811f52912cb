set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "packed_f or conv1d_fwd_dgrad_wgrad or deconv1d_fwd_dgrad_wgrad or bf16_modes or batch_scale" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
B="python bench.py --no-modes --no-cpu-baseline --no-kernel-timer --no-side-workloads --no-host-measure"
for p in fp32 bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$p -o run -- $B --steps 6 --warmup 1 --precision $p > $O/bench_prof_$p.log 2>&1
  python scripts/rocpd_stats.py $O/prof_$p/*results.db $O/kernel_stats_$p.csv 7 > /dev/null 2>&1; rm -rf $O/prof_$p
done
tail -3 $O/tests.log; grep -E "pack_" $O/kernel_stats_fp32.csv $O/kernel_stats_bf16.csv | cut -c1-200
python bench.py --no-cpu-baseline --no-side-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['other_precisions'].items() if k!='note'})"
