set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; rm -rf $O; mkdir -p $O
SHAPE=vanilla11 bash scripts/pmc_sq.sh $O enc0
python scripts/pmc_sq_summary.py $O enc0 > $O/sq.json 2>$O/sq.err
for pass in A B; do f=$(find $O/sq${pass}_enc0 -name "*counter_collection.csv" | head -1); python - "$f" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    if 'edge' in k or 'small' in k:
        per[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, c in per.items():
    print(k, {cn: round(v / n[(k, cn)]) for cn, v in c.items()})
PY
done
