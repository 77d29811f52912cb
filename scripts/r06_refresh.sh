# round 6, after the last change of the HIP sources: profiles first (so that bench.py's traffic /
# pipe-busy figures are measured on the committed sources), then the bench lines — one GPU call.
#   bash scripts/r06_refresh.sh   ->  gpurun_out/r06p/*, gpurun_out/r06f/*  (copied into profiles/ here)
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r06
bash scripts/round_profiles.sh $R > /dev/null 2>&1
O=gpurun_out/${R}p
# the WSEGAN step (BASELINE config 4) gets its own kernel stats and HBM traffic passes too
B="python bench.py --no-modes --no-cpu-baseline --no-kernel-timer --no-side-workloads --no-host-measure"
rocprofv3 --kernel-trace --stats -d $O/prof_ws -o run -- $B --steps 6 --warmup 1 --wsegan > $O/bench_prof_ws.log 2>&1
python scripts/rocpd_stats.py $O/prof_ws/*results.db $O/kernel_stats_wsegan.csv 7 > /dev/null 2>&1; rm -rf $O/prof_ws
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_ws_$c -o run --output-format csv -- $B --steps 2 --warmup 1 --wsegan > $O/pmc_ws_$c.log 2>&1
done
python scripts/pmc_traffic.py $O/pmc_ws_FETCH_SIZE/run_counter_collection.csv $O/pmc_ws_WRITE_SIZE/run_counter_collection.csv 3 $O/pmc_hbm_traffic_wsegan.json '--wsegan (WSEGAN step with --misalign_pair)'
rm -rf $O/pmc_ws_FETCH_SIZE $O/pmc_ws_WRITE_SIZE
python scripts/bench_layers.py --shape vanilla11 --iters 3 > $O/layers_vanilla11.txt 2>&1
P=profiles
cp $O/kernel_stats_fp32.csv $P/${R}_bench_kernel_stats.csv; cp $O/kernel_stats_bf16.csv $P/${R}_bench_kernel_stats_bf16.csv
cp $O/kernel_stats_bf16x3.csv $P/${R}_bench_kernel_stats_bf16x3.csv; cp $O/kernel_stats_vanilla11.csv $P/${R}_bench_kernel_stats_vanilla11.csv
cp $O/kernel_stats_wsegan.csv $P/${R}_bench_kernel_stats_wsegan.csv
cp $O/pmc_hbm_traffic_fp32.json $P/${R}_pmc_hbm_traffic.json; cp $O/pmc_hbm_traffic_bf16.json $P/${R}_pmc_hbm_traffic_bf16.json
cp $O/pmc_hbm_traffic_vanilla11.json $P/${R}_pmc_hbm_traffic_vanilla11.json; cp $O/pmc_hbm_traffic_wsegan.json $P/${R}_pmc_hbm_traffic_wsegan.json
cp $O/sq_counters_fp32.json $P/${R}_sq_counters.json; cp $O/sq_counters_bf16.json $P/${R}_sq_counters_bf16.json
cp $O/layers_fp32.txt $P/${R}_layers.txt; cp $O/layers_bf16.txt $P/${R}_layers_bf16.txt; cp $O/layers_vanilla11.txt $P/${R}_layers_vanilla11.txt
F=gpurun_out/${R}f; mkdir -p $F
cp $P/${R}_*.csv $P/${R}_pmc*.json $P/${R}_sq*.json $P/${R}_layers*.txt $F/ 2>/dev/null
( time python bench.py ) > $F/bench_line.json 2> $F/bench_line.err
python bench.py --wsegan --no-modes --no-side-workloads > $F/bench_line_wsegan.json 2> $F/bench_line_wsegan.err
python bench.py --shape vanilla11 --no-modes --no-cpu-baseline > $F/bench_line_vanilla11.json 2> /dev/null
python bench.py --precision bf16 --device-z --no-modes --no-cpu-baseline --no-side-workloads > $F/bench_line_bf16.json 2> /dev/null
python scripts/train_loop_bench.py --items 30000 --epochs 3 2>/dev/null | tail -1 > $F/train_loop.json
cut -c1-300 $F/bench_line.json; tail -4 $F/bench_line.err; cut -c1-200 $F/bench_line_wsegan.json; cut -c1-200 $F/bench_line_vanilla11.json; cut -c1-200 $F/bench_line_bf16.json; cat $F/train_loop.json
