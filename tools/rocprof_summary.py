#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as a small markdown table.

    python tools/rocprof_summary.py gpurun_out/prof_x/<host>/<pid>_results.db [top_n] > profiles/<name>.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([^(]{0,140})', name)
    return m.group(1).strip()


def main():
    db = sqlite3.connect(sys.argv[1])
    top_n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows = list(db.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    total = sum(r[2] for r in rows)
    print('| kernel | calls | total ms | avg us | % of GPU time |')
    print('|---|---:|---:|---:|---:|')
    for name, calls, dur, avg, pct in rows[:top_n]:
        print('| `%s` | %d | %.2f | %.1f | %.2f |' % (short(name), calls, dur / 1e3, avg, pct))
    print('\nGPU kernel time total: %.1f ms over %d kernels (durations in the db are us).' % (total / 1e3, len(rows)))


if __name__ == '__main__':
    main()
