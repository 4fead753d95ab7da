#!/bin/bash
# Eager two-stream schedule against HIP-graph replay under the runtime's graph knobs, one box, alternating.
run() {  # env-string, extra args
  ms=$(env $1 python bench.py --no-cpu-baseline --no-inference --no-kernel-timing --steps 40 $2 2>/dev/null | python -c "import sys,json; [print('%.3f' % json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")
  echo "$1 $2 : $ms"
}
for rep in 1 2; do
  run "X=0" "--eager"
  run "X=0" "--graph"
  run "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "--graph"
  run "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "--graph"
  run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "--graph"
  run "X=0" "--graph --no-overlap-wgrad"
done
