#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel in a .hip file (compiler remarks; no GPU needed).
#   tools/kernel_resources.sh dmesh2_renderer_amd/csrc/dm2_backward_fast.hip [extra hipcc flags...]
f=$1; shift
cd "$(dirname "$f")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$(basename "$f")" -o /tmp/kres_$$.o 2>&1 |
  grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size|ScratchSize" | sed -E 's/^.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/{n=$0; sub(/.*Function Name: /,"",n); printf "\n%s\n   ", substr(n,1,60); next} {gsub(/^ +/,""); printf "%s | ", $0} END{print ""}'
rm -f /tmp/kres_$$.o
