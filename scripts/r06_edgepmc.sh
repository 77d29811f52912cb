set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; rm -rf $O; mkdir -p $O
bash scripts/pmc_sq.sh $O enc0 dec4
for layer in enc0 dec4; do for pass in A B; do f=$(find $O/sq${pass}_$layer -name "*counter_collection.csv" | head -1); python - "$f" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:48]
    if 'edge' in k or 'small' in k:
        per[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, c in per.items():
    print(k, {cn.replace('SQ_',''): round(v / n[(k, cn)] / 1e6, 2) for cn, v in c.items()})
PY
done; done
