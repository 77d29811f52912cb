set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --no-modes --no-cpu-baseline --no-kernel-timer > $O/pmc_$c.log 2>&1
done
python scripts/pmc_traffic.py $O/pmc_FETCH_SIZE/run_counter_collection.csv $O/pmc_WRITE_SIZE/run_counter_collection.csv 3 $O/pmc_hbm_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
