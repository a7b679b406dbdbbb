#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/r02l
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 120 python tools/host_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02l/host_cost.txt
