# round 4: weight gradients on a side stream (SEGAN_WGRAD_OVERLAP) vs one stream, alternating on one box
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-modes --no-kernel-timer --steps 10"
for p in fp32 bf16; do
  for v in 1 0 1 0; do
    SEGAN_WGRAD_OVERLAP=$v $B --precision $p 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$p overlap=$v', round(d['ms_per_step'], 3))"
  done
done
