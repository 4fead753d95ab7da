#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_libs/old.so vs new.so are copied over the in-tree library in turn.
for i in 1 2 3; do
  for v in old new; do
    cp tools/ab_libs/$v.so margipose_amd/libmargipose_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), round(d['inference']['images_per_sec'],1))"
  done
done
cp tools/ab_libs/new.so margipose_amd/libmargipose_hip.so
