set -u
O=gpurun_out/r03g
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
SEGAN_DETERMINISTIC=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 6 --warmup 1 --no-modes --no-cpu-baseline --no-kernel-timer > $O/bench_prof.log 2>&1
python scripts/rocpd_stats.py $O/prof/*results.db $O/kernel_stats_det.csv 7 > /dev/null 2>&1; rm -rf $O/prof
cut -c1-150 $O/kernel_stats_det.csv | head -24
