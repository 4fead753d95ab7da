#!/bin/bash
# Per-kernel register / scratch / LDS usage of one csrc file (hipcc remarks), e.g. tools/kernel_resources.sh conv
f=${1:-conv}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c margipose_amd/csrc/$f.hip -o /tmp/_res_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|LDS Size" | sed -E 's/.*remark: [^ ]+ //; s/\[-Rpass.*//' | paste - - - - - \
 | sed -E 's/Function Name: _ZN5mpose12_GLOBAL__N_1[0-9]*//; s/ScratchSize \[bytes\/lane\]/Scratch/; s/LDS Size \[bytes\/block\]/LDS/'
