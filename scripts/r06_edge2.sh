set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "long_edge or ragged_length or edge_geometry or first_layer" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
