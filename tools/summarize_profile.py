#!/usr/bin/env python
"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'mpose::', '', name)
    return name[:110]


def main(out):
    for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True) + glob.glob(os.path.join(out, 'tail', '**', '*kernel_stats.csv'), recursive=True):
        print('== kernel stats (%s)' % os.path.relpath(f, out))
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        print('%-112s %8s %12s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'pct'))
        for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:25]:
            print('%-112s %8s %12.3f %10.2f %6.2f' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                      float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
        print('total kernel time ms: %.3f' % (tot / 1e6))
    for sub in ('pmc_sq', 'pmc_lds', 'pmc_fetch', 'pmc_write'):
        files = glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True)
        if not files:
            continue
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for f in files:
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[(k, r['Counter_Name'])] += 1
        print('== PMC pass %s (per-dispatch averages)' % sub)
        for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:12]:
            print(k)
            print('    ' + '  '.join('%s=%.4g' % (c, v / max(cnt[(k, c)], 1)) for c, v in sorted(agg[k].items())))


def traffic_table(out):
    """Per (kernel, grid) averages of FETCH_SIZE / WRITE_SIZE (KB per dispatch, as rocprofv3 reports them)."""
    import json
    tab = defaultdict(lambda: defaultdict(list))
    for sub, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        for f in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == ctr:
                    tab[(short(r['Kernel_Name']), int(r['Grid_Size']))][ctr].append(float(r['Counter_Value']))
    print('== HBM-side traffic per dispatch by (kernel, grid): KB as reported (FETCH_SIZE of 16-B/lane streams is 1/2 of the bytes: MI355X_MICROARCH.md)')
    rows = []
    for (k, gsz), d in tab.items():
        fe = sum(d['FETCH_SIZE']) / max(1, len(d['FETCH_SIZE'])); wr = sum(d['WRITE_SIZE']) / max(1, len(d['WRITE_SIZE']))
        rows.append((fe + wr, k, gsz, len(d['FETCH_SIZE']), fe, wr))
    for tot, k, gsz, n, fe, wr in sorted(rows, reverse=True)[:24]:
        print('%-80s grid %8d n=%4d  FETCH %10.0f KB  WRITE %10.0f KB' % (k[:80], gsz, n, fe, wr))


if __name__ == '__main__':
    main(sys.argv[1])
    traffic_table(sys.argv[1])
