#!/usr/bin/env python3
"""Gap structure of the MFMA stream in a kernel's hottest basic block (from `hipcc -S --cuda-device-only`):
for every v_mfma, how many other instructions precede it and whether it accumulates into the same registers as the
previous MFMA ('s') or different ones ('D').  usage: mfma_gaps.py file.s <mangled-name-substring>"""
import re, sys
src = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*%s\S*):' % re.escape(sys.argv[2]), src, re.M)
start = m.end(); end = src.index('.Lfunc_end', start)
blocks = []; cur = []; lab = 'entry'
for l in src[start:end].split('\n'):
    if re.match(r'^\.LBB\S+:', l):
        blocks.append((lab, cur)); cur = []; lab = l.strip()
    else:
        cur.append(l)
blocks.append((lab, cur))
for lab, b in blocks:
    n = sum('v_mfma' in x for x in b)
    if n > 40:
        out = []; cnt = 0; prev = None; kinds = {}
        for l in b:
            t = l.strip()
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            if t.startswith('v_mfma'):
                dst = re.search(r'v_mfma\S+ (\S+),', t).group(1)
                out.append((cnt, dst == prev)); prev = dst; cnt = 0
            else:
                cnt += 1
                k = t.split()[0].split('_')[0]
                kinds[k] = kinds.get(k, 0) + 1
        print(lab, 'instrs', len([1 for l in b if l.strip() and not l.strip().startswith((';', '.'))]), 'mfma', n, kinds)
        print('  same-acc gaps filled: %d (fillers %d, max %d)   diff-acc gaps filled: %d (fillers %d, max %d)' % (
            sum(1 for c, s in out if s and c), sum(c for c, s in out if s), max([c for c, s in out if s] or [0]),
            sum(1 for c, s in out if not s and c), sum(c for c, s in out if not s), max([c for c, s in out if not s] or [0])))
        print('  ' + ' '.join('%d%s' % (c, 's' if s else 'D') for c, s in out))
