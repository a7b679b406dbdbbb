#!/bin/bash
# the evidence run of round 3: GPU suite, smoke, the driver-exact bench line, rocprofv3 + PMC passes of the same command,
# per-family lines, Brax parity  ->  gpurun_out/r03ev/ (what is cited is copied into profiles/)
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03ev; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver rc=$? lines=$(wc -l < $O/bench_driver_cmd.json)"
timeout 900 tools/profile_gpu.sh r03_driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; cp gpurun_out/prof_r03_driver/summary.txt $O/driver_cmd_rocprofv3_summary.txt
python tools/make_traffic_json.py gpurun_out/prof_r03_driver "profiles/r03_driver_cmd_rocprofv3.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --gpus 1 --steps 20 --warmup 5, round-3 binary)" > $O/traffic_update.txt 2>&1; cp profiles/traffic.json $O/traffic.json
ENVS="pendulum cartpole acrobot mountaincar mountaincar_cont" STEPS=200 tools/bench_all.sh --no-per-call > $O/bench_all_families.txt 2>&1; cat $O/bench_all_families.txt
head -24 $O/driver_cmd_rocprofv3_summary.txt
rm -rf gpurun_out/prof_r03_driver/kt gpurun_out/prof_r03_driver/pmc_fetch gpurun_out/prof_r03_driver/pmc_write
