#!/bin/bash
# Training-step rate with each of the reference's feature extractors (B=32, T=3), one box, back to back.
for st in inceptionv4 resnet18 resnet34 resnet50; do
  timeout 400 python bench.py --stem $st --no-cpu-baseline --no-kernel-timing --steps 10 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$st', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')"
done
