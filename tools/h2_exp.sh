#!/bin/bash
# Builds variants of the library with pieces of conv_h2_k removed (CH_EXP bits, csrc/conv_h.hip) into _ab/ (run HERE), or, with
# `run`, times them on the GPU box (tools/bench_h2.py): what do the DMA, the fragment reads, the MFMAs and the epilogue cost?
cd $(dirname $0)/..
if [ "$1" = run ]; then
  for e in ${CH_LIST:-0 8 16 32 64 128}; do
    echo "CH_EXP=$e"; MPOSE_LIB=_ab/ch_exp$e.so python tools/with_lib.py tools/bench_h2.py 2>&1 | grep -v amdgpu.ids
  done
  exit 0
fi
mkdir -p _ab
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-slp-vectorize"
OBJS=$(ls margipose_amd/csrc/*.o | grep -v "/conv_h.o")
for e in ${CH_LIST:-0 8 16 32 64 128}; do
  ( /opt/rocm/bin/hipcc $F -DCH_EXP=$e $CH_DEFS -c margipose_amd/csrc/conv_h.hip -o _ab/conv_h_exp$e.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/ch_exp$e.so $OBJS _ab/conv_h_exp$e.o ) &
done
wait
ls -la _ab/ch_exp*.so
