#!/bin/bash
export CARL_AMD_NO_BUILD=1
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_brax_invariants.py -m gpu -q -x 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_brax.py -m gpu -q 2>&1 | tail -8
