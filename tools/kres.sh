#!/bin/bash
# tools/kres.sh <tu_name without .hip> <kernel-name substring> — register / scratch / occupancy of the kernels of one translation unit (device-only compile)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -fvisibility=hidden --cuda-device-only -c \
  -Rpass-analysis=kernel-resource-usage $3 -o /tmp/kres.o /root/repo/trace.jl_amd/csrc/$1.hip 2>&1 | sed 's/.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | \
  awk -v pat="$2" '/Function Name/{name=$0; show=(index($0,pat)>0)} show && /VGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs:/{printf "%s | ", $0} show && /LDS Size/{print " <- " name}'
