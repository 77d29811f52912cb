set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "edge_geometry or first_layer or conv1d_fwd_dgrad_wgrad or deconv1d_fwd_dgrad_wgrad or tiny or segan_plus_step or vanilla11" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
python scripts/bench_layers.py --iters 7 --only enc0 2>/dev/null | cut -c1-80
python scripts/bench_layers.py --iters 7 --only dec4 2>/dev/null | cut -c1-80
python scripts/bench_layers.py --shape vanilla11 --iters 7 --only enc0 2>/dev/null | cut -c1-80
python scripts/bench_layers.py --shape vanilla11 --iters 7 --only dec10 2>/dev/null | cut -c1-80
python bench.py --no-cpu-baseline --no-modes --no-host-measure 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step', round(d['ms_per_step'],3), {k:round(v['ms_per_step'],2) for k,v in d['other_workloads'].items() if k!='note'})"
