#!/usr/bin/env python
"""The soft-argmax kernel alone, the same loops as bench.py's tail block (tail_microbench): run it under
   rocprofv3 --kernel-trace --stats  so that profiles/ holds the kernel's average duration next to bench.py's event timing."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda', 0)
out = {'B=2048 fp32 (856 MB/launch)': bench.tail_microbench(dev, 2048, launches=100),
       'B=32 fp32 (configs[2])': bench.tail_microbench(dev, 32, launches=100),
       'B=64 bf16 heatmaps (configs[1])': bench.tail_microbench(dev, 64, bf16_out=True, launches=100),
       'fused with the residual sum, B=2048 fp32': bench.fused_tail_microbench(dev, 2048, launches=100),
       'fused with the residual sum, B=32 fp32 (configs[2])': bench.fused_tail_microbench(dev, 32, launches=100),
       'fused with the residual sum, B=64 bf16 heatmaps (configs[1])': bench.fused_tail_microbench(dev, 64, bf16_out=True, launches=100)}
print(json.dumps(out, indent=1))
