mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_graph_gpu.py -x -q -m gpu -p no:cacheprovider -k "stale" 2>&1 | grep -v Warning | tail -60
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "train_step_vs_oracle or eval_mode_batchnorm or step_switches or every_gradient or two_forwards or overlap_wgrad or retain_graph or eval_forward" 2>&1 | tail -8
bash tools/step_stats.sh r6a 2>&1 | tail -40
