#!/bin/bash
# per-kernel register / scratch / LDS usage of one translation unit, as the compiler reports it
#   scripts/kernel_resources.sh trl_net.hip [extra hipcc flags]
src=$(pwd)/tetris-reinforcement-learning_amd/csrc/$1; shift
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value "$@" \
  -c "$src" -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        if cur: print(cur)
        cur = {"name": v.replace("_ZN12_GLOBAL__N_1", "")}
    else: cur[k.split(" [")[0]] = v
if cur: print(cur)
'
