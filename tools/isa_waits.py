#!/usr/bin/env python3
"""Event stream of one kernel's ISA: barriers, vmcnt waits, branches and labels with the instruction counts between them
(valu / lds / mfma / vmem).  usage: tools/isa_waits.py file.s 'mangled-name-substring' [last N lines]"""
import sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm')][0]
cnt = dict(v=0, l=0, m=0, g=0)
out = []
for l in lines[start:end]:
    t = l.strip().split()
    if not t or t[0].startswith(';'):
        continue
    op = t[0]
    if op == 's_barrier' or (op == 's_waitcnt' and 'vmcnt' in l) or op.startswith('s_cbranch') or op.endswith(':'):
        out.append("   [valu %d lds %d mfma %d vmem %d]" % (cnt['v'], cnt['l'], cnt['m'], cnt['g'])); cnt = dict(v=0, l=0, m=0, g=0)
        out.append(l.strip()[:90])
    elif op.startswith('v_mfma'): cnt['m'] += 1
    elif op.startswith('ds_'): cnt['l'] += 1
    elif op.startswith(('global_', 'buffer_', 'scratch_')): cnt['g'] += 1
    elif op.startswith('v_'): cnt['v'] += 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(out)
print('\n'.join(out[-n:]))
