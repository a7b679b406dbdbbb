#!/bin/bash
# round-2 first GPU call: tests, the driver's exact bench command, rocprofv3 evidence, host costs
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r02a/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver.json 2> gpurun_out/r02a/bench_driver.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02a/bench_driver.json
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err; echo "bench default rc=$?"
timeout 200 python tools/host_cost.py > gpurun_out/r02a/host_cost.txt 2>&1; cat gpurun_out/r02a/host_cost.txt
timeout 400 tools/profile_gpu.sh r02a_pendulum > /dev/null 2>&1; cat gpurun_out/prof_r02a_pendulum/summary.txt
timeout 400 tools/profile_gpu.sh r02a_cartpole --env cartpole --steps 200 --warmup 20 --no-cpu-baseline --no-per-call --also none > /dev/null 2>&1; cat gpurun_out/prof_r02a_cartpole/summary.txt
