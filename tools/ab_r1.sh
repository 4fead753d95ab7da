#!/bin/bash
# A/B of the current build against the round-1 tree (exported to _r1/ by `git archive 08e5a4e | tar -x -C _r1` and built there),
# alternating on ONE box.  Output: gpurun_out/ab_r1.txt
out=gpurun_out/ab_r1.txt; : > $out
for i in 1 2 3; do
  (cd _r1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('r1        ', round(d['value'],1), round(d['ms_per_step'],3), d['inference']['images_per_sec'])") >> $out
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('head      ', round(d['value'],1), round(d['ms_per_step'],3), d['inference']['images_per_sec'])" >> $out
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --graph --no-inference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('head graph', round(d['value'],1), round(d['ms_per_step'],3))" >> $out
done
cat $out
