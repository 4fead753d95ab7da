#!/bin/bash
# Environment switches of the engine under the launch-plan dispatch, interleaved on ONE box: ms per step (bench.py, 30 steps)
run() {
  r=$(env $1 python bench.py --no-cpu-baseline --no-inference --no-kernel-timing --no-configs4 --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.3f  %s' % (d['ms_per_step'], d['config']['step_dispatch'][:28]))")
  echo "$1 : $r"
}
for rep in 1 2; do
  for v in "X=0" "$@"; do run "$v"; done
done
