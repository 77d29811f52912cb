# Round profiles: everything under profiles/rNN_* that comes from rocprofv3 / the layer benchmarks.
# usage (on the GPU box, from the repo root):  bash scripts/round_profiles.sh r03
set -u
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${R}p; rm -rf $O; mkdir -p $O
B="python bench.py --no-modes --no-cpu-baseline --no-kernel-timer --no-side-workloads --no-host-measure"
# per-kernel time of the step, fp32 (the default: deterministic reductions) and the bf16 modes
for p in fp32 bf16 bf16x3; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$p -o run -- $B --steps 6 --warmup 1 --precision $p > $O/bench_prof_$p.log 2>&1
  python scripts/rocpd_stats.py $O/prof_$p/*results.db $O/kernel_stats_$p.csv 7 > /dev/null 2>&1; rm -rf $O/prof_$p
done
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (counters only)
for p in fp32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${p}_$c -o run --output-format csv -- $B --steps 2 --warmup 1 --precision $p > $O/pmc_${p}_$c.log 2>&1
  done
  python scripts/pmc_traffic.py $O/pmc_${p}_FETCH_SIZE/run_counter_collection.csv $O/pmc_${p}_WRITE_SIZE/run_counter_collection.csv 3 $O/pmc_hbm_traffic_$p.json "--precision $p, SEGAN+ default net"
  rm -rf $O/pmc_${p}_FETCH_SIZE $O/pmc_${p}_WRITE_SIZE
done
# the 11-layer stride-2 shape gets its OWN counters (round-4 review, weak 8: its bench line used to
# carry the SEGAN+ profile's traffic / pipe-busy figures): kernel stats + the two PMC passes
rocprofv3 --kernel-trace --stats -d $O/prof_v11 -o run -- $B --steps 6 --warmup 1 --shape vanilla11 > $O/bench_prof_v11.log 2>&1
python scripts/rocpd_stats.py $O/prof_v11/*results.db $O/kernel_stats_vanilla11.csv 7 > /dev/null 2>&1; rm -rf $O/prof_v11
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_v11_$c -o run --output-format csv -- $B --steps 2 --warmup 1 --shape vanilla11 > $O/pmc_v11_$c.log 2>&1
done
python scripts/pmc_traffic.py $O/pmc_v11_FETCH_SIZE/run_counter_collection.csv $O/pmc_v11_WRITE_SIZE/run_counter_collection.csv 3 $O/pmc_hbm_traffic_vanilla11.json '--shape vanilla11 (11-layer stride-2 SEGAN)'
rm -rf $O/pmc_v11_FETCH_SIZE $O/pmc_v11_WRITE_SIZE
# every contraction of the SEGAN+ nets in isolation
python scripts/bench_layers.py --iters 3 > $O/layers_fp32.txt 2>&1
SEGAN_PRECISION=bf16 python scripts/bench_layers.py --iters 3 > $O/layers_bf16.txt 2>&1
# SQ counters of single layers (two passes of 8 counters each)
for p in fp32 bf16; do
  mkdir -p $O/sq_$p; SEGAN_PRECISION=$p bash scripts/pmc_sq.sh $O/sq_$p enc2 dec2 > /dev/null 2>&1
  python scripts/pmc_sq_summary.py $O/sq_$p enc2 dec2 > $O/sq_counters_$p.json 2>/dev/null
  rm -rf $O/sq_$p
done
ls -la $O
