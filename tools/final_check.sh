#!/bin/bash
# Round-end check on one GPU box: full GPU test suite, smoke, default bench line, rocprofv3 profile of the same command.
set -x
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
# the configuration-size gradient cases the default suite skips for time (patch8 same-piece) + the free-running statistics
MPOSE_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_grad_parity_gpu.py -x -q -m gpu -k config_size 2>&1 | tail -3
python tools/collect_gradparity.py r5
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
bash tools/profile.sh r5 > gpurun_out/profile_r5.log 2>&1; tail -5 gpurun_out/profile_r5.log
bash tools/step_stats.sh r5 > gpurun_out/step_families_r5.txt 2>&1; head -3 gpurun_out/step_families_r5.txt
# 2 ranks sharing the GPU through gloo (functional check of the DP bench path)
MPOSE_SINGLE_DEVICE=1 MPOSE_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
# timeline of the default (launch plan, two streams) schedule + the launch-by-launch listing of its last step
bash tools/trace_step.sh r5 > gpurun_out/timeline_r5.txt 2>&1
python tools/step_listing.py $(find gpurun_out/tl_r5/trace -name "*kernel_trace.csv" | head -1) > gpurun_out/step_listing_r5.txt; rm -rf gpurun_out/tl_r5/trace
# the driver's form of the bench
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_driver_form.json 2>/dev/null; cut -c1-300 gpurun_out/bench_driver_form.json
