"""Scan a kernel's ISA (hipcc -S --cuda-device-only output) for the pattern that cost the epilogues of
round 6: a vector-memory LOAD issued after a STORE.  On gfx950 loads and stores retire through one
in-order counter (vmcnt), so the wait for such a load is a wait for every store before it; element by
element (`c[i] += x`, or a per-lane bias load between a tile's stores) that is one store round trip per
element.  Reports, per kernel, the `s_waitcnt vmcnt(0)` and loads that follow the first store in program
text (loop bodies that precede their epilogue in the text are not counted: read the ISA before acting).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -S --cuda-device-only -o /tmp/k.s segan_pytorch_amd/csrc/segan_conv.hip
    python scripts/scan_load_after_store.py /tmp/k.s
"""
import re
import sys

cur = None
res = {}
seen_store = False
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1)
        res[cur] = [0, 0, 0]
        seen_store = False
        continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith('.Lfunc_end'):
        cur = None
        continue
    op = t.split(' ')[0].split('\t')[0]
    if re.match(r'(buffer|global|flat)_store|buffer_atomic|global_atomic', op):
        res[cur][0] += 1
        seen_store = True
    elif re.match(r'(buffer|global|flat)_load', op) and 'lds' not in t:
        if seen_store:
            res[cur][1] += 1
    elif op == 's_waitcnt' and 'vmcnt(0)' in t and seen_store:
        res[cur][2] += 1
for k, (st, ld, w) in sorted(res.items(), key=lambda x: -x[1][2]):
    if st and w:
        print('{:4d} vmcnt(0) and {:4d} loads after the first of {:4d} stores: {}'.format(w, ld, st, k[:120]))
