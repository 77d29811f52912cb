set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "edge_geometry or first_layer or conv1d_fwd_dgrad_wgrad or deconv1d_fwd_dgrad_wgrad or tiny or segan_plus_step or vanilla11" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -2 $O/tests.log
for w in 0 1 0 1; do
  echo "tk64=$w"
  SEGAN_WGRAD_EDGE_TK64=$w python scripts/bench_layers.py --iters 7 --only enc0 2>/dev/null | grep wgrad | cut -c1-80
  SEGAN_WGRAD_EDGE_TK64=$w python scripts/bench_layers.py --iters 7 --only dec4 2>/dev/null | grep wgrad | cut -c1-80
  SEGAN_WGRAD_EDGE_TK64=$w python scripts/bench_layers.py --shape vanilla11 --iters 7 --only enc0 2>/dev/null | grep wgrad | cut -c1-80
  SEGAN_WGRAD_EDGE_TK64=$w python scripts/bench_layers.py --shape vanilla11 --iters 7 --only dec10 2>/dev/null | grep wgrad | cut -c1-80
done
