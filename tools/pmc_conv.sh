#!/bin/bash
# FETCH_SIZE / L2 hit counters of the conv micro-benchmark (own PMC pass, kernel-trace only)
OUT=gpurun_out/pmc_conv_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o pmc -- python tools/bench_conv.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -f csv -d $OUT/l2 -o pmc -- python tools/bench_conv.py > $OUT/l2.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ('fetch', 'l2'):
    agg = collections.defaultdict(list)
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('::')[-1][:40], r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(k, len(v), '%.4g' % (sum(v) / len(v)))
PY
