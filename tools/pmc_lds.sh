#!/bin/bash
# LDS bank-conflict share of the convolution kernels (one PMC pass on one training step) -> gpurun_out/pmc_lds_quick/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds_quick; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $R -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --no-inference > $R/log.txt 2>&1
python - <<'PY'
import csv, glob, collections, re, os
R = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'pmc_lds_quick')
c = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(R + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::|mpose::|void ', '', r['Kernel_Name']).split('(')[0]
        c[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(c.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:14]:
    if v.get('SQ_LDS_IDX_ACTIVE'):
        print('%-44s conflict/active = %.3f' % (k[:44], v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']))
PY
