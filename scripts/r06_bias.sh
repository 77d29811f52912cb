set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "gemm or stft or wgrad or determin or head or linear or fc or pow or tiny or bias or conv1d_fwd or deconv1d_fwd" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
python scripts/bench_layers.py --iters 5 2>/dev/null | grep -E "TOTAL" | cut -c1-80
python bench.py --no-cpu-baseline --no-modes --no-host-measure 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step', round(d['ms_per_step'],3), 'det', round(d.get('ms_per_step_deterministic') or 0,3), d['roofline']['frac'], {k:round(v['ms_per_step'],2) for k,v in d['other_workloads'].items() if k!='note'})"
