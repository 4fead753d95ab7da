#!/bin/bash
# Builds variants of the library with pieces of conv_igemm_k removed (CV_EXP bits, csrc/conv.hip) into _ab/ (run HERE), or, with
# `run`, times them on the GPU box (tools/bench_igemm.py): what do the K loop, the split-K exchange and the epilogue cost?
cd $(dirname $0)/..
if [ "$1" = run ]; then
  for e in ${CV_LIST:-0 1 4 5 2 7}; do
    echo "CV_EXP=$e"; MPOSE_LIB=_ab/cv_exp$e.so python tools/with_lib.py tools/bench_igemm.py 2>&1 | grep -v amdgpu.ids
  done
  exit 0
fi
mkdir -p _ab
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-slp-vectorize"
OBJS=$(ls margipose_amd/csrc/*.o | grep -v "/conv.o")
for e in ${CV_LIST:-0 1 4 5 2 7}; do
  ( /opt/rocm/bin/hipcc $F -DCV_EXP=$e -c margipose_amd/csrc/conv.hip -o _ab/conv_exp$e.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/cv_exp$e.so $OBJS _ab/conv_exp$e.o ) &
done
wait
ls -la _ab/cv_exp*.so
