#!/bin/bash
export CARL_AMD_NO_BUILD=1
mkdir -p gpurun_out/r02i
python tools/brax_parity_percentiles.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02i/brax_parity_percentiles.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
