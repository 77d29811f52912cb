# round 4: start offset between co-resident workgroups (segan_stagger_start) — per-layer rates per offset
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_stagger; mkdir -p $O
for c in 0 15000 30000 60000 100000; do
  SEGAN_STAGGER_CYC=$c timeout 120 python scripts/bench_layers.py --iters 5 > $O/fp32_$c.txt 2>&1
done
for c in 0 8000 16000 32000 64000; do
  SEGAN_PRECISION=bf16 SEGAN_STAGGER_CYC_BF16=$c timeout 120 python scripts/bench_layers.py --iters 5 > $O/bf16_$c.txt 2>&1
done
echo "fp32: TF/s per layer at stagger 0 / 15000 / 30000 / 60000 / 100000 cycles"
paste <(cut -c1-28,60-67 $O/fp32_0.txt) <(cut -c60-67 $O/fp32_15000.txt) <(cut -c60-67 $O/fp32_30000.txt) <(cut -c60-67 $O/fp32_60000.txt) <(cut -c60-67 $O/fp32_100000.txt)
echo "bf16: 0 / 8000 / 16000 / 32000 / 64000"
paste <(cut -c1-28,60-67 $O/bf16_0.txt) <(cut -c60-67 $O/bf16_8000.txt) <(cut -c60-67 $O/bf16_16000.txt) <(cut -c60-67 $O/bf16_32000.txt) <(cut -c60-67 $O/bf16_64000.txt)
