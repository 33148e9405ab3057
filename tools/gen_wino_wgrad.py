# generates cpg_amd/csrc/conv3x3_wino_wgrad.hip
def clob(p): return ', '.join('"a%d"' % (16*p+e) for e in range(16))
mma = ''.join('#define WW_ONE_%d(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[%d:%d], %%0, %%1, a[%d:%d]" : : "v"(A), "v"(B) : %s)\n'
              % (p, 16*p, 16*p+15, 16*p, 16*p+15, clob(p)) for p in range(16))
zero = ''.join('    asm volatile("' + '\\n\\t'.join('v_accvgpr_write_b32 a%d, 0' % (16*p+e) for e in range(16)) + '" : : : %s);\n' % clob(p) for p in range(16))
rd = ''.join('#define WW_RD_%d(m) asm volatile("' % e + '\\n\\t'.join('v_accvgpr_read_b32 %%%d, a%d' % (p, 16*p+e) for p in range(16))
             + '" : ' + ', '.join('"=v"(m[%d])' % p for p in range(16)) + ')\n' for e in range(16))
out_calls = ''.join('        WW_RD_%d(m); out_e(%d, m);\n' % (e, e) for e in range(16))

# slot tables: for k-steps 0..5: slot p -> T micro p (0..13), G loads on k-steps 1, 2
def kstep(ks, cur, nxt):
    lines = []
    for p in range(16):
        work = []
        if ks < 6:
            if p < 14: work.append('t_micro(%d, %d, A%s, B%s)' % (p, ks + 1, nxt, nxt))
            if ks == 1 and p < 14: work.append('g_load(%d, nst_idx)' % p)
            if ks == 2 and p < 14: work.append('g_load(%d, nst_idx)' % (14 + p))
        else:
            if p < 7: work.append('; '.join('w_store(%d)' % (4 * p + q) for q in range(4)))
            else:
                ms = {7: [0, 1], 8: [2, 3], 9: [4], 10: [5, 6], 11: [7, 8], 12: [9, 10], 13: [11], 14: [12], 15: [13]}[p]
                work.append('; '.join('if (more) t_micro(%d, 0, A%s, B%s)' % (m, nxt, nxt) for m in ms))
        lines.append('        WW_ONE_%d(A%s[%d], B%s[%d]); %s; WW_FENCE();\n' % (p, cur, p, cur, p, '; '.join(work) if work else ''))
    return ''.join(lines)

body = ''
sets = ['0', '1']
cur = 0
for half in range(2):
    body += '        {   // stage %s of the pair\n            const bool more = st + %d + 1 < nst; const int nst_idx = st + %d + 1;\n' % ('A' if half == 0 else 'B', half, half)
    if half == 1:
        body += '            if (st + 1 >= nst) break;\n'
    for ks in range(7):
        body += '            // k-step %d\n' % ks + kstep(ks, sets[cur], sets[cur ^ 1]).replace('        WW_', '            WW_')
        cur ^= 1
    body += '            advance_stage();\n        }\n'

src = open('/tmp/wgw_template.hip').read()
src = src.replace('@@MMA@@', mma).replace('@@ZERO@@', zero).replace('@@RD@@', rd).replace('@@OUT@@', out_calls).replace('@@BODY@@', body)
open('/root/repo/cpg_amd/csrc/conv3x3_wino_wgrad.hip', 'w').write(src)
print('ok', len(src))
