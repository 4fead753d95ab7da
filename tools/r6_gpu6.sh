mkdir -p gpurun_out
MPOSE_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "eight_ranks" 2>&1 | grep -v "Warning\|warn" | tail -70
