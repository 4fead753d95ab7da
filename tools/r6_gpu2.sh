mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_h2_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15
timeout 900 python -m pytest tests/test_grad_parity_gpu.py -x -q -m gpu -p no:cacheprovider -k "1-2-auto or 1-8-auto" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_train_graph_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "stale or planned_iterations or train_step_vs_oracle or eval_mode_batchnorm or step_switches or every_gradient or two_forwards or overlap_wgrad" 2>&1 | tail -8
bash tools/ab_sweep.sh "MPOSE_H2_PLANES=0" 2>&1 | tail -6
