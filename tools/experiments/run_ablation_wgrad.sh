#!/bin/bash
for v in base wg_no_split wg_no_loads wg_mfma_only base; do
  echo "== $v"; MPOSE_LIB=margipose_amd/_abl/lib_$v.so timeout 200 python tools/bench_conv.py 2>&1 | grep "^wgrad"
done
