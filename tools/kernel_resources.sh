#!/bin/bash
# VGPRs / spills / occupancy / LDS of every kernel of one HIP source (cross-compiles, no GPU needed):
#   tools/kernel_resources.sh ba_kernels.hip [name-filter]
cd "$(dirname "$0")/../bundle_adjustment_solver_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|Occupancy|LDS Size" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/ {name=$3} /^VGPRs:/ {v=$2} /SGPRs Spill/ {ss=$3} /VGPRs Spill/ {vs=$3} /Occupancy/ {o=$3} /LDS Size/ {printf "%-90s VGPR %3s occ %s spill v%s s%s LDS %s\n", name, v, o, vs, ss, $4}' |
  { if [ -n "$2" ]; then grep "$2"; else cat; fi; }
