#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [grep-pattern]   -- VGPR / scratch / occupancy / LDS of every kernel in a TU
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: //; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' | c++filt | grep -E "$pat"
