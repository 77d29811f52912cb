set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03sq; rm -rf $O; mkdir -p $O
export SEGAN_PRECISION=${1:-bf16}
bash scripts/pmc_sq.sh $O enc2 dec2 > /dev/null 2>&1
python scripts/pmc_sq_summary.py $O enc2 dec2 > $O/sq.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03sq/sq.json'))
for layer,ks in d.items():
    for k,v in ks.items():
        print(layer, k[:70])
        print('   ', {kk: round(vv,3) for kk,vv in v.items() if kk!='raw'})
PY
rm -rf $O/sq?_*/
