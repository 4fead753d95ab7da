#!/bin/bash
# Same-box comparison of the current build with the round-1 tree (exported to _r1/ by `git archive 08e5a4e | tar -x -C _r1`, built
# there): rocprofv3 kernel stats of both (serial schedule, 7 steps), then default-schedule bench lines, alternating.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/abprof; rm -rf $R; mkdir -p $R
(cd _r1 && NO_INF=1 rocprofv3 --kernel-trace --stats -f csv -d $R/r1 -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad > $R/r1.log 2>&1)
rocprofv3 --kernel-trace --stats -f csv -d $R/head -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --no-inference > $R/head.log 2>&1
for i in 1 2; do
  (cd _r1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null > $R/r1_$i.json)
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null > $R/head_$i.json
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-inference --graph 2>/dev/null > $R/head_graph.json
for f in $R/*.json; do python -c "
import sys,json; d=json.loads(open('$f').readline()); i=d.get('inference',{})
print('%-40s %7.1f img/s %7.3f ms   inference %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], i.get('images_per_sec')))"; done | tee $R/summary.txt
python tools/diff_kernel_stats.py $R/r1/b_kernel_stats.csv $R/head/b_kernel_stats.csv 7 | head -40 | tee -a $R/summary.txt
