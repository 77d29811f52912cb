# usage: prof_step.sh <precision> <outdir>: per-kernel stats of the training step (7 steps)
set -u
P=$1; O=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 6 --warmup 1 --no-modes --no-cpu-baseline --no-kernel-timer --precision $P > $O/bench_prof.log 2>&1
python scripts/rocpd_stats.py $O/prof/*results.db $O/kernel_stats_$P.csv 7 > /dev/null 2>&1; rm -rf $O/prof
cut -c1-130 $O/kernel_stats_$P.csv | head -45
