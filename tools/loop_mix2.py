#!/usr/bin/env python3
"""Instruction mix of every innermost loop ("Inner Loop Header" ... backward branch) of one kernel in a hipcc -S listing.
   python tools/loop_mix2.py /tmp/wgrad.s conv_wgrad_rows_kILi2ELi2ELi2ELi2E [-v]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and pat in l and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
def cls(t):
    return ('M' if 'mfma' in t else 'w' if t.startswith('s_waitcnt') else 'a' if 'accvgpr' in t else 'v' if t.startswith('v_') else
            'd' if t.startswith('ds_') else 'b' if t.startswith(('buffer_', 'global_')) else 'B' if t.startswith(('s_cbranch', 's_branch')) else
            's' if t.startswith('s_') else '?')
for i, l in enumerate(body):
    if 'Inner Loop Header' in l:
        lab = l.split(':')[0]
        j = next(k for k in range(i + 1, len(body)) if re.search(r's_c?branch\w* ' + re.escape(lab) + r'\b', body[k]))
        seq = []
        for x in body[i:j + 1]:
            t = x.strip().split(' ')[0]
            if not t or t.startswith('.') or t.startswith(';'):
                continue
            seq.append(cls(t))
        print(lab, len(seq), 'instructions', dict(Counter(seq)))
        if '-v' in sys.argv:
            print(''.join(seq))
