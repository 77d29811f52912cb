"""Summarise the SQ counter passes of scripts/pmc_sq.sh per kernel template:
MFMA-busy share of the SIMD time, VALU / LDS / VMEM instructions per MFMA, wait buckets.
usage: python scripts/pmc_sq_summary.py OUTDIR LAYER [LAYER...] > profiles/rNN_sq_counters.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, cycle-constants table); with 4 SIMDs
per CU, mfma_busy = MFMA_BUSY / (4 * 4 * BUSY_CYCLES per CU-sum) is reported both raw and as
the ratio the guide's derived MfmaUtil uses."""
import collections
import csv
import glob
import json
import sys

out, layers = sys.argv[1], sys.argv[2:]
res = {}
for layer in layers:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for p in 'AB':
        for f in glob.glob('%s/sq%s_%s/**/*counter_collection.csv' % (out, p, layer), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name']
                if 'corr' not in k and 'wgrad' not in k:
                    continue
                per[k][r['Counter_Name']] += float(r['Counter_Value'])
                if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'SQ_INSTS_MFMA'):
                    n[(k, r['Counter_Name'])] += 1
    for k, c in per.items():
        mf = c.get('SQ_INSTS_MFMA', 0.0) or 1.0
        wc = c.get('SQ_WAVE_CYCLES', 0.0) or 1.0
        res.setdefault(layer, {})[k] = {
            'launches': n[(k, 'SQ_WAVE_CYCLES')],
            # issued matrix work per launch: every v_mfma_f32_32x32x16_bf16 is 32768 FLOP, every
            # v_mfma_f32_32x32x2_f32 4096 — against the layer's useful FLOPs (bench_layers: enc2
            # 156.03 GFLOP, dec2 312.06 GFLOP per call) this is the share of the issued matrix work
            # that is padding (the 32nd tap, halo columns, edge tiles)
            'mfma_insts_per_launch': mf / max(1, n[(k, 'SQ_INSTS_MFMA')]),
            'mfma_gflop_issued_per_launch': mf / max(1, n[(k, 'SQ_INSTS_MFMA')]) *
                                            (32768 if 'bf' in k else 4096) / 1e9,
            'mfma_busy_cycles_over_4x_busy_cycles': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4.0 * (c.get('SQ_BUSY_CYCLES', 0) or 1.0)),
            'valu_per_mfma': c.get('SQ_INSTS_VALU', 0) / mf,
            'lds_per_mfma': c.get('SQ_INSTS_LDS', 0) / mf,
            'vmem_per_mfma': c.get('SQ_INSTS_VMEM', 0) / mf,
            'salu_per_mfma': c.get('SQ_INSTS_SALU', 0) / mf,
            'wave_cycles_wait_any': c.get('SQ_WAIT_ANY', 0) / wc,
            'wave_cycles_wait_inst_any': c.get('SQ_WAIT_INST_ANY', 0) / wc,
            'wave_cycles_active_inst_any': c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
            'wave_cycles_wait_inst_lds': c.get('SQ_WAIT_INST_LDS', 0) / (per[k].get('SQ_WAVE_CYCLES', 0) or 1.0),
            'lds_bank_conflict_over_idx_active': c.get('SQ_LDS_BANK_CONFLICT', 0) / (c.get('SQ_LDS_IDX_ACTIVE', 0) or 1.0),
            'raw': dict(c),
        }

def add_summary(res):
    """mfma_pipe_busy per kernel family = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): the first counts
    MFMA-pipe cycles summed over the 1024 SIMDs (64 per fp32 MFMA, 32 per bf16 MFMA: checked against
    SQ_INSTS_MFMA), the second the busy cycles summed over the 32 shader engines (checked against the kernel
    durations), so 32 x SQ_BUSY_CYCLES is the SIMD-cycles of the launch AT THE CLOCK IT RAN AT."""
    fam = {'corr2': 'corr2_kernel', 'wgrad2': 'wgrad2_kernel', 'conv_dgrad_short': 'conv_dgrad_short_kernel',
           'corr_bf2': 'corr_bf2_kernel', 'wgrad_bf2': 'wgrad_bf2_kernel'}
    out = {'note': add_summary.__doc__.replace('\n    ', ' ')}
    for name, key in fam.items():
        v = [k['raw']['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * k['raw']['SQ_BUSY_CYCLES'])
             for layer in res.values() if isinstance(layer, dict) for kn, k in layer.items()
             if key in kn and k['raw'].get('SQ_BUSY_CYCLES')]
        if v:
            out[name + '_mfma_pipe_busy_mean'] = sum(v) / len(v)
    res['_summary'] = out


add_summary(res)
json.dump(res, sys.stdout, indent=1)
