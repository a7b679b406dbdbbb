#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02b/pytest.log
ENVS="pendulum cartpole mountaincar acrobot" tools/bench_all.sh --no-per-call
