#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_brax.py tests/test_gpu_brax_invariants.py tests/test_brax_physics_kat.py tests/test_gpu_mixed_and_multiproc.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3 | cut -c1-300
VARIANTS="product w4" bash tools/r03_gpu_i.sh
timeout 300 python tools/soak_brax.py 2>&1 | grep -v amdgpu.ids | tail -12
