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


# the families bench.py reports (its HIP-event clock brackets the whole library call: + k_c3_pack for fwd / dgrad,
# + k_c3_wgrad_reduce for wgrad); k_pw / k_pw_wgrad serve the 1x1 convs AND the un-masked linear layers.
# (round 5) k_wg3<.., SPLIT = true> -- the tail pieces of a launch -- and k_wg_tail_reduce are helpers of the launch whose main kernel is
# k_wg3<.., false>: they are not counted as launches
FAMS = [('conv_fwd', r'k_c3_fwd<.*>, false(, (true|false))*>|k_conv_fwd|k_wg_fwd<\d, false|k_wg1<false|k_wg3<false(?:, \w+){4}, false>|k_stem_fwd|k_stem2_fwd|k_pw<.*>, false'),
        ('conv_dgrad', r'k_c3_fwd<.*>, true(, (true|false))*>|k_conv_dgrad|k_wg_fwd<\d, true|k_wg1<true|k_wg3<true(?:, \w+){4}, false>|k_c3s2_dgrad|k_pw<.*>, true'),
        ('conv_wgrad', r'k_c3_wgrad<|k_c3_wgrad_smallc|k_conv_wgrad|k_wgw|k_stem2_wgrad|k_pw_wgrad'),
        ('fused BatchNorm / ReLU / pool / PReLU', r'k_bn|k_prelu'),
        ('stock torch elementwise / pooling', r'at::native')]


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
    # the families bench.py reports (its HIP-event clock brackets the whole library call: + k_c3_pack for fwd / dgrad,
    # + k_c3_wgrad_reduce for wgrad)
    # (k_pw / k_pw_wgrad serve the 1x1 convs AND the un-masked linear layers, so they get their own row)
    fams = FAMS
    print('\n| family (main kernels only) | calls | total ms | avg ms |\n|---|---:|---:|---:|')
    for fam, pat in fams:
        sel = [r for r in rows if re.search(pat, r[0])]
        calls, dur = sum(r[1] for r in sel), sum(r[2] for r in sel)
        if calls:
            print('| %s | %d | %.2f | %.3f |' % (fam, calls, dur / 1e3, dur / 1e3 / calls))


if __name__ == '__main__':
    main()
