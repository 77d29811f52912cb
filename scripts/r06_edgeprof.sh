set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; rm -rf $O; mkdir -p $O
for shape in segan_plus vanilla11; do
  rocprofv3 --kernel-trace --stats -d $O/$shape -o t -- python scripts/bench_layers.py --shape $shape --iters 7 --only enc0 > $O/$shape.log 2>&1
  f=$(find $O/$shape -name "*kernel_stats.csv" | head -1)
  echo "== $shape"; head -14 "$f" | cut -c1-160
done
