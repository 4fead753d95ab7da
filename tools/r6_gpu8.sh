timeout 900 python -m pytest tests/test_conv_f16x3_gpu.py tests/test_conv_gpu.py tests/test_stem_gpu.py tests/test_chatterbox_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep "passed\|failed"
timeout 900 python -m pytest tests/test_grad_parity_gpu.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "1-2-inceptionv4 or 1-3-auto or fp16_convolution or eval_forward" 2>&1 | grep "passed\|failed"
bash tools/step_stats.sh r6f 2>&1 | head -5
bash tools/ab_sweep.sh "X=1" 2>&1 | tail -4
