set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bf16" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for w in 0 1 0 1; do
  SEGAN_BF2_ORDER=$w SEGAN_PRECISION=bf16 python scripts/bench_layers.py --iters 5 > $O/layers_o$w.txt 2>&1
  echo "order=$w"; grep -E "fwd|dgrad|TOTAL" $O/layers_o$w.txt | grep -v "enc0\|dec4" | awk '{printf "%s %s %s | ", $1, $2, $5} END {print ""}'
  SEGAN_BF2_ORDER=$w python bench.py --precision bf16 --device-z --no-cpu-baseline --no-modes --no-side-workloads --no-host-measure 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline_wgrad']['frac'],4))"
done
