# A/B of two builds of libsegan_hip (ab/old.so, ab/new.so): accumulation error and step time.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in old new; do
  SEGAN_HIP_LIB=$PWD/ab/$v.so python tests/diag/diag_accum.py 24 > gpurun_out/ab/accum_$v.txt 2>&1
done
B="python bench.py --no-modes --no-cpu-baseline --steps 8 --warmup 2"
for r in 1 2; do for v in old new; do
  SEGAN_HIP_LIB=$PWD/ab/$v.so $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline_wgrad']['frac'])" >> gpurun_out/ab/bench.txt
done; done
cat gpurun_out/ab/accum_old.txt gpurun_out/ab/accum_new.txt gpurun_out/ab/bench.txt
