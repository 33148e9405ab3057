#!/usr/bin/env python3
"""Per kernel of a counter pass (SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA ...): share of the pass's cycles,
MFMA-busy fraction, vector-ALU time (4 cycles per wave64 instruction per SIMD) and their sum -- on gfx950 fp32 MFMA and vector
instructions do not overlap, so "VALU time" is MFMA time lost.

    python tools/pmc_step_table.py gpurun_out/pmc_step/resnet50 > profiles/...
"""
import collections
import csv
import glob
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'^void ', '', k).split('(')[0][:100]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (k, r['Dispatch_Id']) not in seen:
            seen.add((k, r['Dispatch_Id']))
            calls[k] += 1
tot = sum(c['SQ_BUSY_CYCLES'] for c in agg.values()) / 32
print('| kernel | calls | share of cycles | MFMA busy | other VALU instr / MFMA | VALU time | sum | LDS instr / MFMA |')
print('|---|---:|---:|---:|---:|---:|---:|---:|')
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_BUSY_CYCLES']):
    cyc = c['SQ_BUSY_CYCLES'] / 32
    if cyc / tot < 0.002:
        continue
    mf = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc
    nm = c['SQ_INSTS_MFMA']
    nv = c['SQ_INSTS_VALU'] - nm                      # SQ_INSTS_VALU counts the MFMA instructions too: the OTHER vector instructions
    va = nv / 1024 * 4 / cyc
    print('| `%s` | %d | %.3f | %.3f | %s | %.3f | %.3f | %s |' % (k, calls[k], cyc / tot, mf, '%.2f' % (nv / nm) if nm else '-', va, mf + va,
                                                              '%.2f' % (c['SQ_INSTS_LDS'] / nm) if nm else '-'))
