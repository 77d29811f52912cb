# round 4: the bench lines committed under profiles/ (run AFTER profiles/r04_pmc_hbm_traffic.json and
# r04_sq_counters.json are in place: bench.py reads its `traffic` / pipe-busy figures from them)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --wsegan --no-modes > $O/bench_line_wsegan.json 2> $O/bench_line_wsegan.err
python bench.py --shape vanilla11 --no-modes --no-cpu-baseline > $O/bench_line_vanilla11.json 2> /dev/null
python scripts/train_loop_bench.py 2>/dev/null | tail -1 > $O/train_loop.json
cut -c1-400 $O/bench_line.json; cut -c1-300 $O/bench_line_wsegan.json; cut -c1-300 $O/bench_line_vanilla11.json; cat $O/train_loop.json
