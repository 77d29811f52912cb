"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
separately, as MI355X_MICROARCH.md prescribes) -> profiles/r01_pmc_hbm_traffic.json, the file
bench.py reads for `roofline.traffic`.
usage: python scripts/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv STEPS OUT.json [WORKLOAD]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
steps = int(sys.argv[3])
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    kernels[k] = {'launches': max(len(f), len(w)),
                  'fetch_kb_avg': sum(f) / len(f) if f else 0.0,
                  'write_kb_avg': sum(w) / len(w) if w else 0.0}
out = {
    'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps {} --warmup 1 '
            '{}, B=300; units KB per launch as reported; gfx950 FETCH_SIZE '
            'under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md HBM section)'.format(
                steps - 1, sys.argv[5] if len(sys.argv) > 5 else 'the SEGAN+ default net'),
    'workload': sys.argv[5] if len(sys.argv) > 5 else 'segan_plus',
    'steps_profiled': steps,
    'csrc_sha': __import__('bench').csrc_sha(),      # the HIP sources these numbers were measured on
    'per_step_fetch_gb_raw': sum(sum(v) for v in fetch.values()) / 1e6 / steps * 1.024,
    'per_step_write_gb': sum(sum(v) for v in write.values()) / 1e6 / steps * 1.024,
    'kernels': kernels,
}
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print('fetch GB/step raw %.2f  write GB/step %.2f' % (out['per_step_fetch_gb_raw'], out['per_step_write_gb']))
