#!/bin/bash
for v in base fix_no_loop fix_tap_const fix_no_masks fix_no_store fix_no_exchange base; do
  echo "== $v"; MPOSE_LIB=margipose_amd/_abl/lib_$v.so timeout 200 python tools/bench_conv.py 2>&1 | grep "^conv"
done
