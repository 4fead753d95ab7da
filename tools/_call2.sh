set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_graph_gpu.py -x -q -m gpu 2>&1 | tail -15
for mode in "" "--eager" "" "--eager"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs4 --no-inference --no-kernel-timing $mode 2> gpurun_out/bench_ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('MODE[$mode]', round(d['ms_per_step'], 3), d['config']['step_dispatch'][:60], d['config']['final_loss'])
"
  tail -3 gpurun_out/bench_ab.err
done
