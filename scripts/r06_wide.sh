set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv1d_fwd_dgrad_wgrad or deconv1d_fwd_dgrad_wgrad or stride2 or edge_geometry" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for w in 0 1 0 1; do
  SEGAN_T_WIDE=$w python scripts/bench_layers.py --shape vanilla11 --iters 5 --only enc > $O/layers_w$w.txt 2>&1
  echo "wide=$w"; grep -E "enc[1-3] dgrad" $O/layers_w$w.txt | cut -c1-80
  SEGAN_T_WIDE=$w python scripts/bench_layers.py --shape vanilla11 --iters 5 --only dec > $O/layersd_w$w.txt 2>&1
  grep -E "dec[789] fwd" $O/layersd_w$w.txt | cut -c1-80
  SEGAN_T_WIDE=$w python bench.py --shape vanilla11 --no-cpu-baseline --no-modes --no-host-measure 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
done
