#!/bin/bash
# SQ counter passes (MFMA busy, VALU per MFMA, LDS / wait buckets) over the per-layer bench of
# the contraction kernels.  usage: [SHAPE=vanilla11] scripts/pmc_sq.sh OUTDIR LAYER [LAYER...]   (e.g. enc2 dec2)
# Counters only (no --sys-trace etc.): two passes of 8 SQ counters each, as the 8 SQ slots allow.
set -u
out=$1; shift
export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
for layer in "$@"; do
  for pass in A B; do
    c=$A; [ $pass = B ] && c=$B
    rocprofv3 --pmc $c --kernel-trace -d $out/sq${pass}_$layer -o run --output-format csv -- \
      python scripts/bench_layers.py --iters 1 --only $layer ${SHAPE:+--shape $SHAPE} > $out/sq${pass}_$layer.log 2>&1
  done
done
