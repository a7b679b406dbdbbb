#!/bin/bash
# the evidence run of a round: GPU suite, the driver-exact bench line, rocprofv3 + PMC passes of the same command,
# per-family line, SQ counters, Brax parity percentiles  ->  gpurun_out/evidence/ (copy what is cited into profiles/)
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/evidence; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver rc=$?"
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> /dev/null
timeout 500 tools/profile_gpu.sh evidence_driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; cp gpurun_out/prof_evidence_driver/summary.txt $O/driver_cmd_rocprofv3_summary.txt
ENVS="pendulum cartpole acrobot mountaincar mountaincar_cont" tools/bench_all.sh > $O/bench_all_families.txt 2>&1; cat $O/bench_all_families.txt
tools/pmc_sq.sh evidence_acrobot --env acrobot --steps 40 --warmup 5 --no-cpu-baseline --no-per-call --also none > $O/acrobot_sq_counters.txt 2>&1
timeout 200 python tools/brax_parity_percentiles.py 2>&1 | grep -v amdgpu.ids > $O/brax_parity_percentiles.txt
head -22 $O/driver_cmd_rocprofv3_summary.txt
