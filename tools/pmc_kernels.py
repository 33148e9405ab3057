#!/usr/bin/env python3
"""HBM bytes per launch of every kernel of a short command, from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate passes, kernel
trace only): bytes = 2 x FETCH_SIZE + WRITE_SIZE in KiB units (profiles/r05_fetch_calib.md), averaged per dispatch, with the dispatch's
duration from the kernel trace of the same pass.

    python tools/pmc_kernels.py gpurun_out/<dir with FETCH_SIZE/ and WRITE_SIZE/> [name-regex]"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return re.match(r'([^(]{0,110})', re.sub(r'^void ', '', name)).group(1).strip()


def main():
    root, pat = sys.argv[1], re.compile(sys.argv[2] if len(sys.argv) > 2 else '.')
    val = {c: collections.defaultdict(list) for c in ('FETCH_SIZE', 'WRITE_SIZE')}
    dur = collections.defaultdict(list)
    for c in val:
        for f in glob.glob(os.path.join(root, c, '**', '*counter_collection.csv'), recursive=True):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == c:
                    per[(r['Dispatch_Id'], r['Kernel_Name'])] += float(r['Counter_Value']) * 1024.0
            for (_, k), v in per.items():
                val[c][short(k)].append(v)
        for f in glob.glob(os.path.join(root, c, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[short(r['Kernel_Name'])].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    print('| kernel | launches | avg us (under the counters) | FETCH_SIZE MB | WRITE_SIZE MB | HBM MB = 2 x fetch + write | TB/s |\n|---|---:|---:|---:|---:|---:|---:|')
    for k in sorted(val['FETCH_SIZE'], key=lambda k: -sum(val['FETCH_SIZE'][k])):
        if not pat.search(k):
            continue
        f = sum(val['FETCH_SIZE'][k]) / max(1, len(val['FETCH_SIZE'][k]))
        w = sum(val['WRITE_SIZE'][k]) / max(1, len(val['WRITE_SIZE'][k]))
        us = sum(dur[k]) / max(1, len(dur[k]))
        print('| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.2f |' % (k, len(val['FETCH_SIZE'][k]), us, f / 1e6, w / 1e6, (2 * f + w) / 1e6, (2 * f + w) / us / 1e6))


if __name__ == '__main__':
    main()
