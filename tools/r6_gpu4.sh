mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=25 > gpurun_out/r6_suite4.txt 2>&1; tail -45 gpurun_out/r6_suite4.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6_bench4.json 2> gpurun_out/r6_bench4.err; tail -c 1500 gpurun_out/r6_bench4.json
