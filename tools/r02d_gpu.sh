#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r02d/pytest.log
ENVS="pendulum cartpole acrobot mountaincar mountaincar_cont" tools/bench_all.sh
tools/pmc_sq.sh r02d_cartpole --env cartpole --steps 40 --warmup 5 --no-cpu-baseline --no-per-call --also none
tools/pmc_sq.sh r02d_acrobot --env acrobot --steps 40 --warmup 5 --no-cpu-baseline --no-per-call --also none
