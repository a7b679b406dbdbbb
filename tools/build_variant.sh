#!/bin/bash
# one experimental build of the library -> gpurun_in/libcarl_<name>.so (git-ignored; travels to the GPU box; loaded with
# CARL_AMD_LIB_PATH).  Only carl_brax.hip is recompiled with the extra flags; carl_amd.o is the product's.
#   tools/build_variant.sh <name> [-D...]
name=$1; shift
mkdir -p gpurun_in /tmp/variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
  -c carl_amd/csrc/carl_brax.hip -o /tmp/variant_$name/carl_brax.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc carl_amd/lib/obj/carl_amd.o /tmp/variant_$name/carl_brax.o -o gpurun_in/libcarl_$name.so
ls -la gpurun_in/libcarl_$name.so
