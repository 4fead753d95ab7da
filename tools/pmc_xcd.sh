#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the convolution kernels with and without conv_igemm_k's XCD-contiguous tile order (MPOSE_IGEMM_XCD).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference --steps 1 --warmup 1"
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    MPOSE_IGEMM_XCD=$v rocprofv3 --kernel-trace --pmc $c -f csv -d gpurun_out/pmc_xcd/$v/$c -o pmc -- $BENCH > gpurun_out/pmc_xcd/log_$v_$c.txt 2>&1
  done
done
python - <<'PY'
import csv, glob, re, collections
for v in '01':
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in glob.glob('gpurun_out/pmc_xcd/%s/%s/**/*counter_collection.csv' % (v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                n = re.sub(r'\(anonymous namespace\)::|mpose::|void ', '', r['Kernel_Name']).split('(')[0]
                if 'conv_igemm_k' in n or 'conv_h2r' in n:
                    ctr[n][r['Counter_Name']].append(float(r['Counter_Value']))
    print('MPOSE_IGEMM_XCD=%s' % v)
    for n, d in sorted(ctr.items()):
        fe = sum(d['FETCH_SIZE']) / max(1, len(d['FETCH_SIZE'])); wr = sum(d['WRITE_SIZE']) / max(1, len(d['WRITE_SIZE']))
        print('  %-46s n=%3d  2xFETCH %8.1f MB  WRITE %8.1f MB  sum %8.1f MB' % (n, len(d['FETCH_SIZE']), 2 * fe / 1024, wr / 1024, (2 * fe + wr) / 1024))
PY
