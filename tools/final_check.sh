#!/bin/bash
# Round-end check on one GPU box: full GPU test suite, smoke, the MPOSE_LONG_TESTS cases, default bench line, rocprofv3 profile of the same
# command (+ PMC passes), step families, two-rank gloo bench, timeline + launch listing of the planned step, the driver's bench form, A/B.
TAG=${1:-r6}
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 -p no:cacheprovider > gpurun_out/final_suite.txt 2>&1; tail -4 gpurun_out/final_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
# the cases the default suite skips for time: configuration-size patch8 same-piece + the free-running statistics, the other feature
# extractors' train steps, the five-stage InceptionV4 fp16 case, T = 2 train step, eight gloo ranks on the one GPU
MPOSE_LONG_TESTS=1 timeout 1800 python -m pytest tests/test_grad_parity_gpu.py -x -q -m gpu -k "config_size or 1-8-auto" -p no:cacheprovider > gpurun_out/final_long_config.txt 2>&1; tail -3 gpurun_out/final_long_config.txt
MPOSE_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_stem_gpu.py -q -m gpu -p no:cacheprovider -k "eight_ranks or (stem_train_step and not resnet34) or (fp16_convolution and 5-384-inceptionv4) or (eval_mode_batchnorm and inceptionv4) or (train_step_vs_oracle and 2-2) or five_stage_model_at_384 or (stem_eval_forward and (resnet18 or resnet50)) or inference_config_batch64" > gpurun_out/final_long_rest.txt 2>&1; tail -3 gpurun_out/final_long_rest.txt
python tools/collect_gradparity.py $TAG
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
bash tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -5 gpurun_out/profile_$TAG.log
bash tools/step_stats.sh $TAG > gpurun_out/step_families_$TAG.txt 2>&1; head -3 gpurun_out/step_families_$TAG.txt
# 2 ranks sharing the GPU through gloo (functional check of the DP bench path)
MPOSE_SINGLE_DEVICE=1 MPOSE_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-inference 2>&1 | tail -1 | cut -c1-400
# timeline of the default (launch plan, two streams) schedule + the launch-by-launch listing of its last step
bash tools/trace_step.sh $TAG > gpurun_out/timeline_$TAG.txt 2>&1
python tools/step_listing.py $(find gpurun_out/tl_$TAG/trace -name "*kernel_trace.csv" | head -1) > gpurun_out/step_listing_$TAG.txt; rm -rf gpurun_out/tl_$TAG/trace
# the driver's form of the bench
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_driver_form.json 2>/dev/null; cut -c1-300 gpurun_out/bench_driver_form.json
# same-box A/B of the round's change: round 5's backward for the H2 blocks against planes end to end
bash tools/ab_sweep.sh "MPOSE_H2_PLANES=0" > gpurun_out/ab_planes_$TAG.txt 2>&1; cat gpurun_out/ab_planes_$TAG.txt
# keep what travels back small
find gpurun_out/prof_$TAG gpurun_out/stats_$TAG -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh gpurun_out | tail -1
