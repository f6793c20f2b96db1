#!/bin/bash
# kernel resource usage of one translation unit: tools/kres.sh tu_trace3c.hip k_trace3c [-Dflags...]
cd "$(dirname "$0")/../trace.jl_amd/csrc"
tu=$1; pat=$2; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -c $tu -o /tmp/kres.o "$@" -Rpass-analysis=kernel-resource-usage ${KRES_SCHED--mllvm -amdgpu-sched-strategy=max-ilp} --offload-device-only 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{name=$3; next} /VGPRs:|ScratchSize|Occupancy|Spill/{a[name]=a[name] " " $0 ";"} END{for(k in a) print k, a[k]}' | grep "$pat" | sed -E 's/_ZN2thL?[0-9]*//; s/EEvNS_[^ ]* / /' | sort
