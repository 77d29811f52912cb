# A/B of two builds of libsegan_hip (ab/old.so, ab/new.so; SEGAN_HIP_LIB selects): step time of
# bench.py, alternating, two rounds.  Optional first argument: a pytest -k expression run on new.so.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/bench.txt
if [ -n "${1:-}" ]; then
  SEGAN_HIP_LIB=$PWD/ab/new.so timeout 45 python -m pytest tests/test_gpu_model.py -q -x -k "$1" 2>&1 | grep -E "passed|failed|rror" | tail -3
fi
B="python bench.py --no-modes --no-cpu-baseline --steps 8 --warmup 2"
for v in new old new; do
  SEGAN_HIP_LIB=$PWD/ab/$v.so $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline_wgrad']['frac'])" >> gpurun_out/ab/bench.txt
done
cat gpurun_out/ab/bench.txt
