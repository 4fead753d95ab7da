#!/usr/bin/env python
"""Timeline of a training step from a rocprofv3 --kernel-trace CSV: wall time between the SGD kernels of consecutive steps,
busy time per queue, time at least one kernel / both queues are running, the sum of idle gaps on the busiest queue, and the
kernel families by time.   python tools/timeline.py <..._kernel_trace.csv> [n_last_steps]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ends = [r['e'] for r in rows if 'sgd_step_k' in r['Kernel_Name']]
assert len(ends) > nlast, 'not enough steps in the trace'
t0, t1 = ends[-nlast - 1], ends[-1]
sel = [r for r in rows if r['s'] >= t0 and r['e'] <= t1]
wall = (t1 - t0) / nlast / 1e6


def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


qs = collections.defaultdict(list)
for r in sel:
    qs[r['Queue_Id']].append((r['s'], r['e']))
print('wall %.2f ms/step over the last %d steps, %d launches/step' % (wall, nlast, len(sel) / nlast))
busy = {q: sum(e - s for s, e in iv) / nlast / 1e6 for q, iv in qs.items()}
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print('  queue %s: busy %.2f ms/step (%d launches/step), union %.2f' % (q, b, len(qs[q]) / nlast, union(qs[q]) / nlast / 1e6))
allu = union([iv for v in qs.values() for iv in v]) / nlast / 1e6
print('  some kernel running: %.2f ms/step -> GPU idle %.2f ms/step; sum of kernel durations %.2f' % (allu, wall - allu, sum(busy.values())))
fam = collections.defaultdict(float)
for r in sel:
    n = re.sub(r'\(anonymous namespace\)::|mpose::|void ', '', r['Kernel_Name']).split('(')[0].split('<')[0][:40]
    fam[n] += (r['e'] - r['s']) / nlast / 1e6
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:14]:
    print('    %-40s %6.2f ms' % (k, v))
# gaps on the main queue (the one with most launches)
mq = max(qs, key=lambda q: len(qs[q]))
iv = sorted(qs[mq]); gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
print('  main queue %s: sum of gaps %.2f ms/step, median gap %.2f us, gaps > 5 us: %d/step (%.2f ms)' % (
    mq, sum(g for g in gaps if g > 0) / nlast / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3, sum(1 for g in gaps if g > 5000) / nlast,
    sum(g for g in gaps if g > 5000) / nlast / 1e6))
# forward / backward phases of the last step (the backward begins with the first loss-backward kernel)
last0 = ends[-2]
st_ = [r for r in rows if r['s'] >= last0 and r['e'] <= ends[-1]]
bwd0 = min((r['s'] for r in st_ if 'stage_loss_bwd_k' in r['Kernel_Name'] or 'average_loss_bwd_k' in r['Kernel_Name']), default=None)
if bwd0 is not None:
    for name, lo, hi in (('forward + loss', last0, bwd0), ('backward + update', bwd0, ends[-1])):
        ph = [r for r in st_ if lo <= r['s'] < hi]
        u = union([(r['s'], min(r['e'], hi)) for r in ph])
        small = sum(r['e'] - r['s'] for r in ph if r['e'] - r['s'] < 12000)
        print('  %-18s wall %.2f ms, some kernel running %.2f, idle %.2f; kernel-time sum %.2f; launches %d (%d under 12 us: %.2f ms)' % (
            name, (hi - lo) / 1e6, u / 1e6, (hi - lo - u) / 1e6, sum(r['e'] - r['s'] for r in ph) / 1e6, len(ph),
            sum(1 for r in ph if r['e'] - r['s'] < 12000), small / 1e6))
