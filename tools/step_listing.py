#!/usr/bin/env python
"""Launch-by-launch listing of the LAST training step of a rocprofv3 --kernel-trace CSV (sgd_step_k to sgd_step_k): start offset,
queue, duration, gap to the previous launch of the same queue, grid, kernel.   python tools/step_listing.py <..._kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ends = [r['e'] for r in rows if 'sgd_step_k' in r['Kernel_Name']]
t0, t1 = ends[-2], ends[-1]
sel = [r for r in rows if r['s'] >= t0 and r['e'] <= t1]
qids = sorted(set(r['Queue_Id'] for r in sel), key=lambda q: -sum(1 for r in sel if r['Queue_Id'] == q))
last_end = {}
print('# step %.3f ms, %d launches; columns: start_us queue dur_us gap_us grid wg kernel' % ((t1 - t0) / 1e6, len(sel)))
for r in sel:
    q = qids.index(r['Queue_Id'])
    gap = (r['s'] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r['e']
    name = re.sub(r'\(anonymous namespace\)::|mpose::|void ', '', r['Kernel_Name'])
    name = re.sub(r'\(.*$', '', name)[:90]
    grid = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
    wg = r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))
    print('%9.1f q%d %8.1f %7.1f %8s %5s %s' % ((r['s'] - t0) / 1e3, q, (r['e'] - r['s']) / 1e3, gap, grid, wg, name))
