#!/bin/bash
for v in base rowg_base mfma_only rowg_mfma_only fix_no_loop rowg_no_loop rowg_no_b rowg_no_apath rowg_no_aload rowg_no_frag base rowg_base; do
  echo "== $v"; MPOSE_LIB=margipose_amd/_abl/lib_$v.so timeout 200 python tools/bench_conv.py 2>&1 | grep "^conv"
done
