# Round 6: stride-2 pair-store epilogue — parity (full GPU suite, timed), then the 11-layer shape.
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/tests_full.log 2>&1
echo "tests rc=$?" >> $O/tests_full.log
python scripts/bench_layers.py --shape vanilla11 --iters 3 > $O/layers_vanilla11.txt 2>&1
python bench.py --shape vanilla11 --no-cpu-baseline --no-modes > $O/bench_vanilla11.log 2>&1
tail -c 3000 $O/tests_full.log; grep -E "enc[1-6] dgrad|dec[0-9] fwd|TOTAL" $O/layers_vanilla11.txt | cut -c1-80; tail -c 300 $O/bench_vanilla11.log
