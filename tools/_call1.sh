set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden_direct_gpu.py -x -q -m gpu -k stem_fixture 2>&1 | tail -15
timeout 600 python -m pytest tests/test_train_helpers.py -x -q -m gpu -k checkpoint 2>&1 | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "bench_launches or train_step_vs_oracle" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_stem_gpu.py -x -q -m gpu -k "resnet34" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_tail_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/tail_variants.sh 2>&1 | tee gpurun_out/tail_variants.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r5_a.json 2> gpurun_out/bench_r5_a.err; cut -c1-300 gpurun_out/bench_r5_a.json
bash tools/ab_graph.sh 2>&1 | tee gpurun_out/ab_graph.log
