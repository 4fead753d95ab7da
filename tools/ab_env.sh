#!/bin/bash
# Interleaved A/B of the training step on ONE box: tools/ab_env.sh "ENV_A=1" "ENV_B=1" [reps] [bench.py args]  -> ms per step of every run
A="$1"; Bv="$2"; R=${3:-3}; shift 3
for i in $(seq $R); do
  for v in "$A" "$Bv"; do
    ms=$(env $v python bench.py --no-cpu-baseline --no-inference --no-kernel-timing --steps 40 "$@" 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")
    echo "$v : $ms"
  done
done
