# round 5, after the last change of the HIP sources: profiles first (so that bench.py's traffic /
# pipe-busy figures are measured on the committed sources), then the bench lines — one GPU call.
#   bash scripts/r05_refresh.sh   ->  gpurun_out/r05p/*, gpurun_out/r05f/*  (copy into profiles/ afterwards)
set -u
cd $GRAFT_REPO_ROOT
bash scripts/round_profiles.sh r05 > /dev/null 2>&1
O=gpurun_out/r05p
cp $O/kernel_stats_fp32.csv profiles/r05_bench_kernel_stats.csv; cp $O/kernel_stats_bf16.csv profiles/r05_bench_kernel_stats_bf16.csv
cp $O/kernel_stats_bf16x3.csv profiles/r05_bench_kernel_stats_bf16x3.csv; cp $O/kernel_stats_vanilla11.csv profiles/r05_bench_kernel_stats_vanilla11.csv
cp $O/pmc_hbm_traffic_fp32.json profiles/r05_pmc_hbm_traffic.json; cp $O/pmc_hbm_traffic_bf16.json profiles/r05_pmc_hbm_traffic_bf16.json
cp $O/pmc_hbm_traffic_vanilla11.json profiles/r05_pmc_hbm_traffic_vanilla11.json
cp $O/sq_counters_fp32.json profiles/r05_sq_counters.json; cp $O/sq_counters_bf16.json profiles/r05_sq_counters_bf16.json
cp $O/layers_fp32.txt profiles/r05_layers.txt; cp $O/layers_bf16.txt profiles/r05_layers_bf16.txt
F=gpurun_out/r05f; mkdir -p $F
python bench.py > $F/bench_line.json 2> $F/bench_line.err
python bench.py --wsegan --no-modes > $F/bench_line_wsegan.json 2> $F/bench_line_wsegan.err
python bench.py --shape vanilla11 --no-modes --no-cpu-baseline > $F/bench_line_vanilla11.json 2> /dev/null
python scripts/train_loop_bench.py --items 30000 --epochs 3 2>/dev/null | tail -1 > $F/train_loop.json
cut -c1-400 $F/bench_line.json; cut -c1-300 $F/bench_line_wsegan.json; cut -c1-300 $F/bench_line_vanilla11.json; cat $F/train_loop.json
