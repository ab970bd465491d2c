#!/usr/bin/env python3
"""Static instruction budget of a kernel between '; DM2_MARK x' comments in hipcc -save-temps ISA."""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
kern = next(l.split(":")[0] for l in open(path) if l.startswith("_Z") and kern in l.split(":")[0])
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
cur = 'PROLOGUE'; order = [cur]
cnt = collections.defaultdict(collections.Counter)
for l in lines[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r'; DM2_MARK (\S+)', t)
    if m:
        cur = m.group(1)
        if cur not in order: order.append(cur)
        continue
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0]
    cls = ('valu' if op.startswith('v_') else 'salu' if op.startswith('s_') and not op.startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_branch', 's_nop'))
           else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'branch' if 'branch' in op else 'wait' if op.startswith(('s_waitcnt', 's_nop')) else 'barrier' if op.startswith('s_barrier') else 'other')
    cnt[cur][cls] += 1
    if op.startswith('scratch_'): cnt[cur]['scratch'] += 1
    for key in ('v_cndmask', 'v_div_', 'v_rcp', 'v_cmp', 'v_mov', 'dpp', 'v_lshl', 'v_lshr', 'v_and', 'v_or', 'v_bcnt'):
        if key in t.split(';')[0] and (op.startswith(key) or key == 'dpp'): cnt[cur][key] += 1
cols = ['valu', 'salu', 'lds', 'vmem', 'branch', 'barrier', 'wait', 'scratch', 'v_cndmask', 'v_cmp', 'v_mov', 'dpp', 'v_div_', 'v_rcp']
print('%-12s' % 'segment' + ''.join('%10s' % c for c in cols))
tot = collections.Counter()
for seg in order:
    print('%-12s' % seg + ''.join('%10d' % cnt[seg][c] for c in cols)); tot.update(cnt[seg])
print('%-12s' % 'TOTAL' + ''.join('%10d' % tot[c] for c in cols))
