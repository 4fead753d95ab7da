#!/bin/bash
# rocprofv3 kernel stats of the current build only (serial schedule, 7 steps) -> gpurun_out/prof_head/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/prof_head; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --stats -f csv -d $R -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --no-inference $@ > $R/log.txt 2>&1
