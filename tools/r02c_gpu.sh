#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 1500 python -m pytest tests -m gpu -q -rs -k "acrobot or config3 or mixed or process_group" > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r02c/pytest.log
ENVS="acrobot cartpole" tools/bench_all.sh --no-per-call
