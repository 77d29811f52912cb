# round 4, after a change of the HIP sources: profiles first (so that bench.py's traffic / pipe-busy
# figures are measured on the committed sources), then the bench lines — one GPU call
set -u
cd $GRAFT_REPO_ROOT
bash scripts/round_profiles.sh r04 > /dev/null 2>&1
O=gpurun_out/r04p
timeout 120 python scripts/clock_under_load.py > $O/clock_under_load.json 2>/dev/null
cp $O/kernel_stats_fp32.csv profiles/r04_bench_kernel_stats.csv; cp $O/kernel_stats_bf16.csv profiles/r04_bench_kernel_stats_bf16.csv
cp $O/kernel_stats_bf16x3.csv profiles/r04_bench_kernel_stats_bf16x3.csv
cp $O/pmc_hbm_traffic_fp32.json profiles/r04_pmc_hbm_traffic.json; cp $O/pmc_hbm_traffic_bf16.json profiles/r04_pmc_hbm_traffic_bf16.json
cp $O/sq_counters_fp32.json profiles/r04_sq_counters.json; cp $O/sq_counters_bf16.json profiles/r04_sq_counters_bf16.json
cp $O/layers_fp32.txt profiles/r04_layers.txt; cp $O/layers_bf16.txt profiles/r04_layers_bf16.txt
cp $O/clock_under_load.json profiles/r04_clock_under_load.json
bash scripts/r04_final.sh
