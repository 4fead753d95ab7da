#!/bin/bash
# per-launch time of the dominant conv shapes vs batch (workgroup rounds): is the per-round fixed cost a bandwidth burst?
for b in 1 2 5 10 11 21 32; do echo "== B $b"; B=$b timeout 200 python tools/bench_conv.py 2>&1 | grep "^conv.*pro=0"; done
