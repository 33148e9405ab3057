#!/usr/bin/env python3
"""Compile one csrc/*.hip for gfx950 and print per-kernel VGPR/AGPR/occupancy/LDS/spill (hipcc
-Rpass-analysis=kernel-resource-usage).   python tools/kernel_resources.py cpg_amd/csrc/conv3x3.hip"""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    if 'error' in line:
        print(line)
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        name = re.sub(r'\(.*', '', name).replace('void ', '')
        cur = {'name': name}
        rows.append(cur)
        continue
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
                     ('lds', r'LDS Size \[bytes/block\]: (\d+)'), ('spill', r'VGPRs Spill: (\d+)'), ('sgpr', r' SGPRs: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print('%-78s %5s %5s %4s %7s %5s' % ('kernel', 'VGPR', 'AGPR', 'occ', 'LDS', 'spill'))
for r in rows:
    print('%-78s %5s %5s %4s %7s %5s' % (r['name'][:78], r.get('vgpr'), r.get('agpr'), r.get('occ'), r.get('lds'), r.get('spill')))
