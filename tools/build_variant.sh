#!/bin/bash
# A measurement / A-B variant of the library next to the product build (cross-compiles here, travels to the GPU box in
# gpurun_in/):   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DCARL_BRAX_PROFILE]
#   -> gpurun_in/libcarl_<name>.so     (load it with CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_<name>.so CARL_AMD_NO_BUILD=1)
# Same flags as carl_amd/build.py.  Only carl_brax.hip is recompiled when BRAX_ONLY=1 (the classic unit's object is reused).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_in; obj=/tmp/carl_variant_$name; mkdir -p "$out" "$obj"
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -fno-slp-vectorize"
/opt/rocm/bin/hipcc $common "$@" -c "$root/carl_amd/csrc/carl_brax.hip" -o "$obj/carl_brax.o" &
if [ "$BRAX_ONLY" = 1 ] && [ -f "$root/carl_amd/lib/obj/carl_amd.o" ]; then cp "$root/carl_amd/lib/obj/carl_amd.o" "$obj/carl_amd.o"
else /opt/rocm/bin/hipcc $common "$@" -c "$root/carl_amd/csrc/carl_amd.hip" -o "$obj/carl_amd.o" & fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc "$obj/carl_amd.o" "$obj/carl_brax.o" -o "$out/libcarl_$name.so"
echo "$out/libcarl_$name.so"
