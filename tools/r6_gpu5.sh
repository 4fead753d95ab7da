mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_h2_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "fp16_convolution or every_gradient" 2>&1 | tail -8
MPOSE_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_stem_gpu.py -q -m gpu -p no:cacheprovider -k "eight_ranks or (stem_train_step and not resnet34) or (fp16_convolution and 5-384-inceptionv4) or (eval_mode_batchnorm and inceptionv4)" 2>&1 | tail -8
bash tools/step_stats.sh r6b 2>&1 | tail -36
