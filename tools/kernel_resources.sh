#!/bin/bash
# Per-kernel register / scratch usage of one csrc file (hipcc remarks), e.g. tools/kernel_resources.sh conv
f=${1:-conv}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -c margipose_amd/csrc/$f.hip -o /tmp/_res_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r'remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)',l)
    if not m: continue
    k,v=m.groups()
    if k=='Function Name':
        cur={'name':v}
    else:
        cur[k.split()[0]]=v
        if k.startswith('Occupancy') or k.startswith('Scratch') and 'done' not in cur:
            pass
    if k.startswith('ScratchSize'):
        print('%-70s VGPR %4s AGPR %4s scratch %s' % (re.sub(r'_ZN5mpose12_GLOBAL__N_1\d+','',cur['name'])[:70], cur.get('VGPRs'), cur.get('AGPRs'), v))
"
