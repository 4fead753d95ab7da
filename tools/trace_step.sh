#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT schedule (weight gradients on the side stream) + tools/timeline.py; run on the GPU box: bash tools/trace_step.sh TAG [bench args]
TAG=${1:-x}; shift
OUT=gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -f csv -d $OUT/trace -o bench -- python bench.py --no-cpu-baseline --no-kernel-timing --no-inference --no-configs4 --steps 6 --warmup 2 "$@" > $OUT/bench_trace.log 2>&1
python tools/timeline.py $(find $OUT/trace -name '*kernel_trace.csv' | head -1) 3
