#!/bin/bash
# Soft-argmax kernel variants at the cache-defeating sizes (MPOSE_TAIL_VARIANT = rows per workgroup + 16 * NT bits; bit 0 stores, bit 1 loads)
for rep in 1 2; do
for v in 1 17 49 2 18 50; do
  MPOSE_TAIL_VARIANT=$v python tools/bench_tail.py --fwd-only 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('variant %3s  B=%4d F=%2d %-15s %7.1f us  %6.0f GB/s  %.3f' % (d['variant'], d['B'], d['F'], d['dtype'], d['us'], d['GBps'], d['frac_of_8TBps']))
"
done
done
