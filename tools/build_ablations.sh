#!/bin/bash
# profiling-only variant libraries (CARL_EXP_* macros) -> gpurun_in/ (git-ignored, travels to the GPU box)
mkdir -p gpurun_in
for v in "NO_DRAIN:-DCARL_EXP_NO_DRAIN" "NO_LOADER:-DCARL_EXP_NO_LOADER" "NO_BOTH:-DCARL_EXP_NO_DRAIN -DCARL_EXP_NO_LOADER" \
         "NO_SINK:-DCARL_EXP_NO_OBS_STORE -DCARL_EXP_NO_REWARD_STORE -DCARL_EXP_NO_FLAG_STORES" \
         "NO_ALL:-DCARL_EXP_NO_OBS_STORE -DCARL_EXP_NO_REWARD_STORE -DCARL_EXP_NO_FLAG_STORES -DCARL_EXP_NO_DRAIN -DCARL_EXP_NO_LOADER"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DCARL_ABLATION $flags $EXTRA carl_amd/csrc/carl_amd.hip carl_amd/csrc/carl_brax.hip \
    -o gpurun_in/libcarl_$name.so 2>&1 | grep -E "error" 
done
ls gpurun_in
