#!/bin/bash
export CARL_AMD_NO_BUILD=1
python tools/diag_cheetah_nan.py 2>&1 | grep -v "amdgpu.ids"
echo "== F64RELE"; CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_BX_F64RELE.so python tools/brax_parity_percentiles.py ant humanoid hopper walker2d pusher humanoidstandup 2>&1 | grep -v "amdgpu.ids"
