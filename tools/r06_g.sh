#!/bin/bash
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1 CARL_PARITY_DETAIL=1
O=$PWD/gpurun_out/r06g; mkdir -p $O
timeout 300 python tools/diag_humanoid_reset_step.py 2>&1 | grep -v amdgpu | head -3
timeout 900 python tools/brax_parity_long.py 16384 300 humanoid ant 2>&1 | grep -v amdgpu | tee $O/parity_long.txt
timeout 600 python tools/brax_parity_percentiles.py 2>&1 | grep -v amdgpu | tee $O/percentiles.txt
timeout 900 python -m pytest tests/test_gpu_brax.py tests/test_gpu_brax_invariants.py -m gpu -x -q 2>&1 | tail -3
