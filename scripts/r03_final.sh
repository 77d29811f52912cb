set -u
cd $GRAFT_REPO_ROOT
bash scripts/round_profiles.sh r03 > /dev/null 2>&1
O=gpurun_out/r03p
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --wsegan --no-modes > $O/bench_line_wsegan.json 2> $O/bench_line_wsegan.err
python bench.py --shape vanilla11 --no-modes --no-cpu-baseline > $O/bench_line_vanilla11.json 2> /dev/null
python scripts/train_loop_bench.py 2>/dev/null | tail -1 > $O/train_loop.json
cut -c1-400 $O/bench_line.json; cut -c1-300 $O/bench_line_wsegan.json; cut -c1-300 $O/bench_line_vanilla11.json; cat $O/train_loop.json
