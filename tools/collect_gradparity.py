#!/usr/bin/env python
"""profiles/<tag>_gradient_parity.json from the summaries tests/test_grad_parity_gpu.py (and test_model_gpu's eval-mode test) write
under gpurun_out/ : per case the statistics only (no per-tensor lists).   python tools/collect_gradparity.py r2"""
import glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r2'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {'_source': 'python -m pytest tests/test_grad_parity_gpu.py tests/test_model_gpu.py -m gpu on an MI355X (gpurun); relative L2 error per '
                  'parameter tensor against the float64 oracle; masked_* = oracle forced onto the ReLU piece the GPU used, free_* / config_* = '
                  'every implementation on its own piece; all cases run the one engine configuration that ships (three fp16 products; the '
                  'regular 128-channel blocks on producer-split planes end to end)'}
for f in sorted(glob.glob(os.path.join(root, 'gpurun_out', 'gradparity_*.json'))):
    d = json.load(open(f))
    out[os.path.basename(f)[len('gradparity_'):-5]] = {k: v for k, v in d.items() if k not in ('per_key', 'worst_gpu')}
json.dump(out, open(os.path.join(root, 'profiles', tag + '_gradient_parity.json'), 'w'), indent=1)
print(len(out) - 1, 'cases')
