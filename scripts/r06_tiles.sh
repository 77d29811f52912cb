# Round 6: the small-row tile variants — kernel parity, then the 11-layer shape per layer and as a step.
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv1d_fwd_dgrad_wgrad or deconv1d_fwd_dgrad_wgrad or stride2 or edge_geometry or batch_scale or short_rows" > $O/tests_kernels.log 2>&1
echo "tests rc=$?" >> $O/tests_kernels.log
python scripts/bench_layers.py --shape vanilla11 --iters 3 > $O/layers_vanilla11.txt 2>&1
python bench.py --shape vanilla11 --no-cpu-baseline --no-modes > $O/bench_vanilla11.log 2>&1
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "vanilla11 or stride2" > $O/tests_model.log 2>&1
echo "tests rc=$?" >> $O/tests_model.log
tail -c 1500 $O/tests_kernels.log; tail -c 400 $O/tests_model.log; grep -E "enc[1-4] |dec[789] |TOTAL" $O/layers_vanilla11.txt | cut -c1-80; tail -c 600 $O/bench_vanilla11.log
