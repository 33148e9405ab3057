#!/usr/bin/env python3
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
# drop everything before the last big gap-free region start: use all
busy = sum(e - s for _, s, e in rows) / 1e6
wall = (rows[-1][2] - rows[0][1]) / 1e6
print('kernels %d  busy %.1f ms  wall %.1f ms  idle %.1f%%' % (len(rows), busy, wall, 100 * (1 - busy / wall)))
gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, rows[i][0][:60], rows[i + 1][0][:60]) for i in range(len(rows) - 1))
print('largest gaps (us):')
for g in gaps[-12:]:
    print('  %.0f  after %s  before %s' % g)
print('gap histogram (us):', {b: sum(1 for g in gaps if lo <= g[0] < hi) for b, (lo, hi) in
      {'<5': (-1e9, 5), '5-20': (5, 20), '20-100': (20, 100), '100-1000': (100, 1000), '>1000': (1000, 1e12)}.items()})
print('total gap ms by bucket:', {b: round(sum(g[0] for g in gaps if lo <= g[0] < hi) / 1e3, 1) for b, (lo, hi) in
      {'<5': (-1e9, 5), '5-20': (5, 20), '20-100': (20, 100), '100-1000': (100, 1000), '>1000': (1000, 1e12)}.items()})
fam = collections.defaultdict(float)
for n, s, e in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    key = n.split('<')[0].split('(')[0][:40]
    fam[key] += (e - s) / 1e6
for k, v in sorted(fam.items(), key=lambda x: -x[1])[:22]:
    print('  %-42s %9.1f ms' % (k, v))
