# usage: prof_layers.sh <precision> <outdir> [bench_layers args]: per-kernel stats of the layer benchmarks
set -u
P=$1; O=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
SEGAN_PRECISION=$P rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/bench_layers.py --iters 3 "$@" > $O/layers.txt 2>&1
python scripts/rocpd_stats.py $O/prof/*results.db $O/kernel_stats.csv 1 > /dev/null 2>&1; rm -rf $O/prof
head -12 $O/kernel_stats.csv | cut -c1-200
