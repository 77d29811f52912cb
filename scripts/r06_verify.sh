# Round 6: what the driver runs at round end, in its order — the GPU suite (timed), smoke(), the default bench.
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 ) > $O/tests_full.log 2>&1
echo "tests rc=$?" >> $O/tests_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -c 1800 $O/tests_full.log; tail -2 $O/smoke.log
