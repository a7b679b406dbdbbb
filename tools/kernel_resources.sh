#!/bin/bash
# VGPRs / scratch / LDS / occupancy of every kernel of one translation unit (cross-compiles, no GPU needed):
#   tools/kernel_resources.sh carl_amd/csrc/carl_brax.hip [-fno-slp-vectorize ...]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/error|warning:/ {print} /Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)} /SGPRs:/ {sg=$(NF-1)} /LDS Size/ {printf "%s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s lds %s\n", name, v, a, sg, sc, o, $(NF-1)}' |
  while read -r n rest; do d=$(echo "$n" | c++filt 2>/dev/null | sed 's/(.*//; s/void //; s/carl:://g' | cut -c1-64); printf "%-64s %s\n" "$d" "$rest"; done
