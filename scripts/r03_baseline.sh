set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
for p in bf16 bf16x3; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$p -o run -- python bench.py --steps 6 --warmup 1 --no-modes --no-cpu-baseline --no-kernel-timer --precision $p > $O/bench_prof_$p.log 2>&1
  python scripts/rocpd_stats.py $O/prof_$p/*results.db $O/kernel_stats_$p.csv 7 > /dev/null 2>&1; rm -rf $O/prof_$p
done
SEGAN_PRECISION=bf16 python scripts/bench_layers.py --iters 3 > $O/layers_bf16.txt 2>&1
SEGAN_PRECISION=bf16x3 python scripts/bench_layers.py --iters 3 > $O/layers_bf16x3.txt 2>&1
SEGAN_DETERMINISTIC=1 python bench.py --steps 10 --warmup 3 --no-modes --no-cpu-baseline > $O/bench_det.json 2>$O/bench_det.err
python bench.py --steps 10 --warmup 3 --no-modes --no-cpu-baseline > $O/bench_fp32.json 2>$O/bench_fp32.err
tail -2 $O/layers_bf16.txt; cut -c1-300 $O/bench_det.json; cut -c1-300 $O/bench_fp32.json
