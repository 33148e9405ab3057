#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 --pmc counters (csv output of one or more passes) with the derived figures used in
DESIGN.md: MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), where kernel cycles come from
SQ_BUSY_CYCLES / 32 shader engines (the effective clock) -- so the figure is clock independent --, VALU / LDS / SALU
instructions per MFMA, LDS bank-conflict share.

    python tools/pmc_table.py gpurun_out/pmcA gpurun_out/pmcB > profiles/<name>.md
"""
import csv
import collections
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.match(r'([^(]{0,120})', name).group(1).strip()


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values per dispatch
    dur = collections.defaultdict(list)
    for root in sys.argv[1:]:
        for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
            byd = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                byd[(r['Dispatch_Id'], r['Kernel_Name'])][r['Counter_Name']] = byd[(r['Dispatch_Id'], r['Kernel_Name'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            for (_, k), c in byd.items():
                for n, v in c.items():
                    per[short(k)][n].append(v)
        for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    kernels = [k for k in per if re.search(r'k_c3_fwd|k_c3_wgrad<|k_pw<|k_pw_wgrad|k_c3b_fwd|k_c3b_wgrad', k)]
    kernels.sort()
    print('| kernel | avg us | MFMA util | VALU/MFMA | SALU/MFMA | LDS inst/MFMA | LDS bank-conflict share | wait-any share |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    for k in kernels:
        c = {n: sum(v) / len(v) for n, v in per[k].items()}
        mf = c.get('SQ_INSTS_MFMA', 0.0)
        if not mf:
            continue
        cycles = c.get('SQ_BUSY_CYCLES', 0.0) / 32.0
        util = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / cycles if cycles else float('nan')
        g = lambda n: c.get(n, float('nan'))
        print('| `%s` | %.0f | %.1f %% | %.2f | %.2f | %.2f | %.0f %% | %.0f %% |' % (
            k, sum(dur[k]) / max(1, len(dur[k])), 100 * util, g('SQ_INSTS_VALU') / mf, g('SQ_INSTS_SALU') / mf, g('SQ_INSTS_LDS') / mf,
            100 * g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE') if g('SQ_LDS_IDX_ACTIVE') else float('nan'),
            100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') else float('nan')))


if __name__ == '__main__':
    main()
