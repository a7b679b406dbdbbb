#!/bin/bash
# Build the HIP library of another commit into gpurun_in/libcarl_<NAME>.so (for A/B runs on one box: tools/ab_probe.sh,
# tools/ab_family.sh, tools/ab_brax.sh):   tools/build_ref.sh <git ref> [NAME=PREV]
set -e
REF=${1:?git ref}; NAME=${2:-PREV}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$ROOT" archive "$REF" carl_amd/csrc include | tar -x -C "$T"
cd "$T"
for f in carl_amd carl_brax; do
  extra=""; [ $f = carl_brax ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function $extra -c carl_amd/csrc/$f.hip -o $f.o &
done
wait
mkdir -p "$ROOT/gpurun_in"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc carl_amd.o carl_brax.o -o "$ROOT/gpurun_in/libcarl_$NAME.so"
rm -rf "$T"
echo "$ROOT/gpurun_in/libcarl_$NAME.so  <- $REF"
