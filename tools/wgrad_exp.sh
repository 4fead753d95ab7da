#!/bin/bash
# Builds variants of the library with pieces of the weight-gradient K loop removed (WG_EXP bits, csrc/wgrad.hip) into _ab/ (run HERE),
# or, with `run`, times them on the GPU box: what does each part of the loop cost?
cd $(dirname $0)/..
if [ "$1" = run ]; then
  for e in 0 1 2 4 8 3 7 15; do
    echo "WG_EXP=$e"; MPOSE_LIB=_ab/wg_exp$e.so python tools/with_lib.py tools/bench_wgrad.py 2>&1 | grep "pro=0 sc=0"
  done
  exit 0
fi
mkdir -p _ab
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-slp-vectorize"
OBJS=$(ls margipose_amd/csrc/*.o | grep -v wgrad.o)
for e in 0 1 2 4 8 3 7 15; do
  ( /opt/rocm/bin/hipcc $F -DWG_EXP=$e -c margipose_amd/csrc/wgrad.hip -o _ab/wgrad_exp$e.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/wg_exp$e.so $OBJS _ab/wgrad_exp$e.o ) &
done
wait
ls -la _ab/*.so
