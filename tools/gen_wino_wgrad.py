#!/usr/bin/env python3
"""Generates cpg_amd/csrc/conv3x3_wino_wgrad.hip from tools/csrc/wino_wgrad_template.hip: the 16 fixed-AGPR MFMA macros, the accumulator
zero-fill / read-back asm and the 2 x 7 x 16 schedule slots of the main loop (which micro-step of the staging work rides behind which
MFMA).  The generated file is what is compiled and committed; re-run after editing the template or the slot tables.

    python tools/gen_wino_wgrad.py
"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def clob(p): return ', '.join('"a%d"' % (16*p+e) for e in range(16))
mma = ''.join('#define WW_ONE_%d(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[%d:%d], %%0, %%1, a[%d:%d]" : : "v"(A), "v"(B) : %s)\n'
              % (p, 16*p, 16*p+15, 16*p, 16*p+15, clob(p)) for p in range(16))
zero = ''.join('    asm volatile("' + '\\n\\t'.join('v_accvgpr_write_b32 a%d, 0' % (16*p+e) for e in range(16)) + '" : : : %s);\n' % clob(p) for p in range(16))
rd = ''.join('#define WW_RD_%d(m) asm volatile("' % e + '\\n\\t'.join('v_accvgpr_read_b32 %%%d, a%d' % (p, 16*p+e) for p in range(16))
             + '" : ' + ', '.join('"=v"(m[%d])' % p for p in range(16)) + ')\n' for e in range(16))
out_calls = ''.join('        WW_RD_%d(m); out_e(%d, m);\n' % (e, e) for e in range(16))

# slot tables: for k-steps 0..5: slot p -> T micro p (0..13), G loads on k-steps 1, 2
# (narrow = the 14-pixel-wide maps: 48 eight-byte loads / stores per stage instead of 28 sixteen-byte ones)
def kstep(ks, cur, nxt, narrow=False, gs=0, half=0, kk=0):
    """gs > 0: the shared-staging variants (GS = gs): NSI = 16 / gs + 8 + 4 / gs items per wave and stage instead of 28; the x rows of the
    stage in flight live in buffer `half` (xb0 / xb1), the next stage's go into the other one, a workgroup barrier between the stores and
    the next stage's first operand reads.  gs == 4 also shares the input transform (b_micro: 7 micro steps + a barrier per k-step; the
    shared-B buffer of the operands made during body k-step kk, for k-step kk + 1, is (kk + 1) & 1)."""
    lines = []
    xcur, xnxt = ('xb%d' % half, 'xb%d' % (half ^ 1))
    nsi = ((32 // gs + 16) if narrow else (16 // gs + 8 + 4 // gs)) if gs else 0
    first = 16 if narrow else 14                 # staging loads that ride in k-step 1 (the rest in k-step 2)
    per = -(-nsi // 7) if gs else 0
    ts = gs == 4
    bb = (kk + 1) & 1
    for p in range(16):
        work = []
        if ks < 6:
            if ts:
                tm = {0: 'b_micro(0, %d, %d, B%s, %s)', 1: 'b_micro(1, %d, %d, B%s, %s)', 3: 'b_micro(2, %d, %d, B%s, %s)',
                      8: 'b_micro(3, %d, %d, B%s, %s)', 9: 'b_micro(4, %d, %d, B%s, %s)', 10: 'b_micro(5, %d, %d, B%s, %s)',
                      11: 'b_micro(6, %d, %d, B%s, %s)'}
                am = {2: 4, 4: 11, 5: 12, 6: 13}
                if p in tm: work.append(tm[p] % (ks + 1, bb, nxt, xcur))
                if p in am: work.append('t_micro(%d, %d, A%s, B%s, %s)' % (am[p], ks + 1, nxt, nxt, xcur))
                if p == 7: work.append('WW_LDS_BARRIER()')
            elif p < 14: work.append('t_micro(%d, %d, A%s, B%s, %s)' % (p, ks + 1, nxt, nxt, xcur))
            if gs:
                if ks == 1 and p < min(first, nsi): work.append('g_load_s(%d)' % p)
                if ks == 2 and p < nsi - first: work.append('g_load_s(%d)' % (first + p))
            elif narrow:
                if 1 <= ks <= 3: work.append('g_load(%d, nst_idx)' % (16 * (ks - 1) + p))
            else:
                if ks == 1 and p < 14: work.append('g_load(%d, nst_idx)' % p)
                if ks == 2 and p < 14: work.append('g_load(%d, nst_idx)' % (14 + p))
        elif ts:
            # last k-step of a stage: the next stage's rows into the other x buffer (13 items over slots 0-5), stage barrier, then the
            # first operands of the next stage from that buffer
            q6, r6 = divmod(nsi, 6)                # nsi items over slots 0-5
            sl, nxt_i = {}, 0
            for k6 in range(6):
                cnt = q6 + (1 if k6 < r6 else 0)
                sl[k6] = list(range(nxt_i, nxt_i + cnt))
                nxt_i += cnt
            if p in sl: work.append('; '.join('w_store_s(%d, %s)' % (i, xnxt) for i in sl[p]))
            if p == 6: work.append('WW_LDS_BARRIER()')
            if p == 7: work.append('b_micro(0, 0, %d, B%s, %s)' % (bb, nxt, xnxt))
            if p == 8: work.append('b_micro(1, 0, %d, B%s, %s)' % (bb, nxt, xnxt))
            if p == 9: work.append('t_micro(4, 0, A%s, B%s, %s)' % (nxt, nxt, xnxt))
            if p == 10: work.append('b_micro(2, 0, %d, B%s, %s)' % (bb, nxt, xnxt))
            if p == 11: work.append('; '.join('t_micro(%d, 0, A%s, B%s, %s)' % (m, nxt, nxt, xnxt) for m in (11, 12, 13)))
            if p == 12: work.append('WW_LDS_BARRIER()')
            if p == 13: work.append('; '.join('b_micro(%d, 0, %d, B%s, %s)' % (m, bb, nxt, xnxt) for m in (3, 4)))
            if p == 14: work.append('; '.join('b_micro(%d, 0, %d, B%s, %s)' % (m, bb, nxt, xnxt) for m in (5, 6)))
        else:
            if p < 7 and gs:
                items = [per * p + q for q in range(per) if per * p + q < nsi]
                work.append('; '.join('w_store_s(%d, %s)' % (i, xnxt) for i in items))
                if p == 6: work.append('WW_LDS_BARRIER()')
            elif p < 7 and narrow: work.append('; '.join('w_store(%d)' % (7 * p + q) for q in range(7) if 7 * p + q < 48))
            elif p < 7: work.append('; '.join('w_store(%d)' % (4 * p + q) for q in range(4)))
            else:
                ms = {7: [0, 1], 8: [2, 3], 9: [4], 10: [5, 6], 11: [7, 8], 12: [9, 10], 13: [11], 14: [12], 15: [13]}[p]
                # (unconditional: after the last stage these read stale LDS and their results are never used -- a branch here costs
                # register copies where its two paths meet, 5 per k-step)
                work.append('; '.join('t_micro(%d, 0, A%s, B%s, %s)' % (m, nxt, nxt, xnxt) for m in ms))
        lines.append('        WW_ONE_%d(A%s[%d], B%s[%d]); %s; WW_FENCE();\n' % (p, cur, p, cur, p, '; '.join(w for w in work if w) if work else ''))
    return ''.join(lines)

def make_body(narrow, gs=0):
  body = ''
  sets = ['0', '1']
  cur = 0
  for half in range(2):
    body += '        {   // stage %s of the pair\n            const int nst_idx = st + %d + 1;\n            (void)nst_idx;\n' % ('A' if half == 0 else 'B', half)
    if half == 1:
        body += '            if (st + 1 >= nst) break;\n'
    for ks in range(7):
        body += '            // k-step %d\n' % ks + kstep(ks, sets[cur], sets[cur ^ 1], narrow, gs, half, 7 * half + ks).replace('        WW_', '            WW_')
        cur ^= 1
    body += '            advance_stage();\n        }\n'
  return body

body, body_n, body_s4, body_s2, body_ns4 = make_body(False), make_body(True), make_body(False, 4), make_body(False, 2), make_body(True, 4)

src = open(os.path.join(ROOT, 'tools', 'csrc', 'wino_wgrad_template.hip')).read()
src = src.replace('@@MMA@@', mma).replace('@@ZERO@@', zero).replace('@@RD@@', rd).replace('@@OUT@@', out_calls).replace('@@BODY@@', body).replace('@@BODY_N@@', body_n).replace('@@BODY_S4@@', body_s4).replace('@@BODY_S2@@', body_s2).replace('@@BODY_NS4@@', body_ns4)
src = '// GENERATED by tools/gen_wino_wgrad.py from tools/csrc/wino_wgrad_template.hip -- edit those, not this file.\n' + src
open(os.path.join(ROOT, 'cpg_amd', 'csrc', 'conv3x3_wino_wgrad.hip'), 'w').write(src)
print('ok', len(src))
