set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 6 --warmup 1 --no-modes --no-cpu-baseline --no-kernel-timer > $O/bench_prof.log 2>&1
python scripts/rocpd_stats.py $O/prof/*results.db $O/kernel_stats.csv 7 > /dev/null 2>&1; rm -f $O/prof/*.db
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --no-modes --no-cpu-baseline --no-kernel-timer > $O/pmc_$c.log 2>&1
done
python scripts/pmc_traffic.py $O/pmc_FETCH_SIZE/run_counter_collection.csv $O/pmc_WRITE_SIZE/run_counter_collection.csv 3 $O/pmc_hbm_traffic.json
rm -rf $O/pmc_FETCH_SIZE/*kernel_trace* $O/pmc_WRITE_SIZE/*kernel_trace*
python scripts/bench_layers.py --iters 3 > $O/layers.txt 2>&1
python scripts/clock_under_load.py > $O/clock_under_load.json 2> /dev/null
python bench.py --steps 10 --warmup 3 --no-modes --wsegan > $O/bench_wsegan.json 2> /dev/null
python bench.py --steps 10 --warmup 3 --no-modes --shape vanilla11 > $O/bench_vanilla11.json 2> /dev/null
tail -3 $O/layers.txt; cut -c1-200 $O/bench_wsegan.json; cut -c1-200 $O/bench_vanilla11.json
