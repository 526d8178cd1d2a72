#!/usr/bin/env python3
"""Structure of one compiled kernel: labels, branches, wait counts and the number of memory / MFMA
instructions between them.  usage: isa_summary.py file.s mangled-name-substring [--regs]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
if "--regs" in sys.argv:
    for m in re.finditer(r'\.amdhsa_kernel (\S*' + pat + r'\S*)(.*?)\.end_amdhsa_kernel', s, re.S):
        b = m.group(2)
        g = lambda k: re.search(k + r' (\d+)', b).group(1)
        print(m.group(1), 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'scratch', g('private_segment_fixed_size'))
    sys.exit(0)
m = re.search(r'^(\S*' + pat + r'\S*):', s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
out = []; cnt = {}
def flush():
    global cnt
    if cnt: out.append('    ' + ', '.join('%s x%d' % kv for kv in cnt.items())); cnt = {}
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith(';'): continue
    if t.startswith('.'):
        if t.startswith('.LBB'): flush(); out.append(t)
        continue
    op = t.split()[0]
    if op.startswith(('s_waitcnt', 's_cbranch', 's_branch', 's_barrier', 's_endpgm')):
        flush(); out.append('  ' + t.split(';')[0].strip())
    elif op.startswith(('global_', 'v_mfma', 's_load', 'ds_', 'buffer_', 'scratch_')):
        cnt[op] = cnt.get(op, 0) + 1
flush()
print('\n'.join(out))
