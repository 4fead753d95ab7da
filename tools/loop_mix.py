#!/usr/bin/env python3
"""Instruction mix and order of the hottest loop of one kernel in a hipcc -S listing.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o /tmp/conv_dev.s margipose_amd/csrc/conv.hip
   python tools/loop_mix.py /tmp/conv_dev.s conv_igemm_kILi4ELi0ELi2ELb0ELi2E
M = MFMA, v = other VALU, d = LDS, b = buffer/global memory, s = SALU, w = s_waitcnt"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and pat in l and l.rstrip().endswith(')') is False and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        seg = body[labels[mm.group(1)]:i]
        n = sum('v_mfma' in x for x in seg)
        if best is None or n > best[0]:
            best = (n, labels[mm.group(1)], i)
n, a, b = best
seq = []
for x in body[a:b]:
    t = x.strip().split(' ')[0]
    if not t or t.startswith('.') or t.startswith(';'):
        continue
    seq.append('M' if 'mfma' in t else 'w' if t.startswith('s_waitcnt') else 'v' if t.startswith('v_') else 'd' if t.startswith('ds_')
               else 'b' if t.startswith(('buffer_', 'global_')) else 's' if t.startswith('s_') else '?')
print('loop of', len(seq), 'instructions:', dict(Counter(seq)))
print(''.join(seq))
