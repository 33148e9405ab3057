#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/csrc/fetch_calib -> the calibration table (counter KiB x 1024 against the 1 GiB every
kernel touches once per launch).   python tools/fetch_calib_table.py gpurun_out/calib > profiles/r05_fetch_calib.md"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
BYTES = float(1 << 30)


def table(sub, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                k = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0]
                agg[k].append(float(r['Counter_Value']) * 1024.0)
    return agg


print('# Round 5: FETCH_SIZE / WRITE_SIZE of rocprofv3 on gfx950 against known byte counts, per access width\n')
print('`tools/csrc/fetch_calib.hip`: every kernel touches each byte of a 1 GiB buffer (4 x the Infinity Cache) exactly once per launch, coalesced, in the')
print('width its name says; FETCH_SIZE and WRITE_SIZE collected in separate `rocprofv3 --pmc <counter> --kernel-trace` passes (no other trace domain), counter')
print('values in KiB x 1024; three launches each.  `k_read_rows_b64` is the Winograd kernels\' patch-row pattern: a wave reads 32 column pairs (8 B per lane,')
print('256 contiguous bytes), the neighbouring 256 bytes belong to another wave.\n')
for sub, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    print('| kernel | %s per launch | / bytes touched | factor to multiply the counter with |' % counter)
    print('|---|---:|---:|---:|')
    for k, v in sorted(table(sub, counter).items()):
        m = sum(v) / len(v)
        if m < 1e6:
            continue
        print('| `%s` | %.1f MiB | %.3f | %.2f |' % (k, m / (1 << 20), m / BYTES, BYTES / m))
    print()
