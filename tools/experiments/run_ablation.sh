#!/bin/bash
# on the GPU box: time every ablation build on the dominant conv shapes (tools/bench_conv.py, conv lines only)
for v in base no_b no_frag no_apath no_aload no_split b_only frag_only mfma_only base; do
  echo "== $v"; MPOSE_LIB=margipose_amd/_abl/lib_$v.so timeout 200 python tools/bench_conv.py 2>&1 | grep "^conv"
done
