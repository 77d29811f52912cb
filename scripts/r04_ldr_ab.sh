# A/B of the loader-wave form of corr_bf2_kernel (round 4): per-layer bf16 rates of three builds / switches
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_ldr; mkdir -p $O
SEGAN_HIP_LIB=$PWD/ab/ldr4.so SEGAN_PRECISION=bf16 SEGAN_BF2_LDR=0 timeout 120 python scripts/bench_layers.py --iters 5 > $O/layers_base.txt 2>&1
SEGAN_HIP_LIB=$PWD/ab/ldr4.so SEGAN_PRECISION=bf16 SEGAN_BF2_LDR=1 timeout 120 python scripts/bench_layers.py --iters 5 > $O/layers_ldr4.txt 2>&1
SEGAN_HIP_LIB=$PWD/ab/ldr3.so SEGAN_PRECISION=bf16 SEGAN_BF2_LDR=1 timeout 120 python scripts/bench_layers.py --iters 5 > $O/layers_ldr3.txt 2>&1
paste <(cut -c1-28,52-70 $O/layers_base.txt) <(cut -c52-70 $O/layers_ldr4.txt) <(cut -c52-70 $O/layers_ldr3.txt) | grep -v wgrad
SEGAN_HIP_LIB=$PWD/ab/ldr4.so SEGAN_BF2_LDR=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "bf16" 2>&1 | tail -3
