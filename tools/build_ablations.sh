#!/bin/bash
# profiling-only variant libraries (CARL_EXP_* macros) -> gpurun_in/ (git-ignored, travels to the GPU box)
mkdir -p gpurun_in
for v in "NO_DRAIN:-DCARL_EXP_NO_DRAIN" "NO_LOADER:-DCARL_EXP_NO_LOADER" "NO_BOTH:-DCARL_EXP_NO_DRAIN -DCARL_EXP_NO_LOADER" \
         "NO_SINK:-DCARL_EXP_NO_OBS_STORE -DCARL_EXP_NO_REWARD_STORE -DCARL_EXP_NO_FLAG_STORES" \
         "NO_DRAW:-DCARL_EXP_NO_DRAW" \
         "NO_ALL:-DCARL_EXP_NO_OBS_STORE -DCARL_EXP_NO_REWARD_STORE -DCARL_EXP_NO_FLAG_STORES -DCARL_EXP_NO_DRAIN -DCARL_EXP_NO_LOADER"; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DCARL_ABLATION $flags $EXTRA -c carl_amd/csrc/carl_amd.hip -o /tmp/abl_$name.o 2>&1 | grep -E "error"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc /tmp/abl_$name.o carl_amd/lib/obj/carl_brax.o -o gpurun_in/libcarl_$name.so ) &
done
wait
ls gpurun_in
