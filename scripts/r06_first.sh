# Round 6, first GPU call: the new tests, where the host time goes, the 11-layer shape per layer,
# and the default bench line with its `other_workloads` block.
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
python -m pytest tests/test_pcm_shard.py tests/test_gpu_dist.py "tests/test_gpu_model.py::test_constant_skip_matches_reference_and_is_never_trained" "tests/test_gpu_model.py::test_mse_reg_loss_matches_reference" "tests/test_gpu_model.py::test_wsegan_literal_train_matches_reference" -m gpu -x -q > $O/tests_new.log 2>&1
echo "tests rc=$?" >> $O/tests_new.log
for a in "--shape segan_plus" "--shape vanilla11" "--wsegan"; do
  n=$(echo $a | tr -d ' -'); python scripts/host_profile.py $a > $O/host_$n.txt 2>&1
done
python scripts/host_profile.py --wsegan --no-prefetch --cprofile 0 > $O/host_wsegan_noprefetch.txt 2>&1
python scripts/bench_layers.py --shape vanilla11 --iters 3 --verbose > $O/layers_vanilla11.txt 2>&1
( time python bench.py ) > $O/bench_default.log 2>&1
tail -c 600 $O/tests_new.log; head -c 1500 $O/host_shapevanilla11.txt
