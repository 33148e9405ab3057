#!/usr/bin/env python3
"""Counters of tools/pmc_wino.sh per kernel, normalised to the kernel's own cycles (SQ_BUSY_CYCLES / 32)."""
import collections
import csv
import glob
import sys

KEEP = ('k_wg_', 'k_c3_fwd', 'k_pw')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for root in sys.argv[1:]:
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            per[(r['Kernel_Name'], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
        for (k, _), c in per.items():
            for n, v in c.items():
                agg[k][n].append(v)
for k, c in agg.items():
    if not any(t in k for t in KEEP):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k[:110])
    cyc = m.get('SQ_BUSY_CYCLES', 0) / 32
    if cyc:
        print('  kernel cycles %.3e   MFMA util %.3f' % (cyc, m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc))
    for n, v in sorted(m.items()):
        print('  %-28s %.4e' % (n, v))
