#!/bin/bash
for v in base ks1 ks2 ks4 base; do
  echo "== $v"; MPOSE_LIB=margipose_amd/_abl/lib_$v.so timeout 200 python tools/bench_conv.py 2>&1 | grep "^conv"
done
