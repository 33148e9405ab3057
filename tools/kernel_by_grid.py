#!/usr/bin/env python3
"""Per grid size: calls and average duration of the kernels whose name matches a pattern, from a rocprofv3 --kernel-trace rocpd database
(the `kernels` view): tells the launches of one kernel instance apart by layer.

    python tools/kernel_by_grid.py gpurun_out/prof_x/run_results.db 'k_pw<.*true, false, true>'
"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2])
agg = collections.defaultdict(list)
for name, gx, wx, start, end in db.execute('select name, grid_x, workgroup_x, start, end from kernels'):
    short = re.sub(r'\(anonymous namespace\)::', '', name)
    if pat.search(short):
        agg[(short.split('(')[0][:90], gx // max(1, wx))].append((end - start) / 1e3)
print('| kernel | blocks | calls | avg us |')
print('|---|---:|---:|---:|')
for (k, blocks), v in sorted(agg.items()):
    print('| `%s` | %d | %d | %.1f |' % (k, blocks, len(v), sum(v) / len(v)))
