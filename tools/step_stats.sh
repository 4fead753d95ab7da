#!/bin/bash
# Kernel-trace stats of N training steps (serial schedule) grouped into families; run on the GPU box.   bash tools/step_stats.sh TAG
TAG=${1:-r3}
OUT=gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o bench -- python bench.py --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference --steps 5 --warmup 2 > $OUT/bench_trace.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 7.0
fam = collections.defaultdict(lambda: [0, 0.0])
def family(n):
    n = re.sub(r'\(anonymous namespace\)::|mpose::|void ', '', n)
    for k in ('conv_igemm_k', 'conv_wgrad_rows_k', 'conv_wgrad_k', 'conv_planes_k'):
        if k in n: return k
    return n.split('(')[0].split('<')[0][:40]
for r in rows:
    k = family(r['Name']); fam[k][0] += int(r['Calls']); fam[k][1] += float(r['TotalDurationNs'])
tot = sum(v[1] for v in fam.values())
print('per step (%d steps incl. warm-up): %.2f ms kernel time, %d launches' % (steps, tot / 1e6 / steps, sum(v[0] for v in fam.values()) / steps))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:32]:
    print('%-42s %7.1f launches  %8.3f ms  avg %7.1f us' % (k, v[0] / steps, v[1] / 1e6 / steps, v[1] / v[0] / 1e3))
PY
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
