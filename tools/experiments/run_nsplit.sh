#!/bin/bash
for n in 1 2 3 4 5 6 7 8 9 10 12; do echo "== NSPLIT $n"; NSPLIT=$n timeout 200 python tools/bench_conv.py 2>&1 | grep "^wgrad.*pro=0"; done
