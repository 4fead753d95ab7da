for X in 0 12000 30000; do echo "extra_lds=$X"; MPOSE_DEBUG_EXTRA_LDS=$X python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value']); print({k:v for k,v in d['kernel_time_breakdown_ms_per_step'].items() if k.startswith('conv')})
"; done
