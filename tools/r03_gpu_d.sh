#!/bin/bash
# round 3, GPU call D: re-run of the suite, per-call period vs rocprofv3 trace, 2-process plumbing, rccl world-1 bench
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -rs > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-300; grep -n "^____\|^E  " $O/pytest_gpu.log | head -20
timeout 200 python tools/per_call_period.py 2>&1 | grep -v amdgpu.ids | tee $O/per_call_period.txt
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/pc_trace -o pc -- python $R/tools/per_call_period.py > $R/$O/per_call_period_under_rocprof.txt 2>&1); grep -v "amdgpu.ids\|^W2026" $O/per_call_period_under_rocprof.txt | tail -3 | tee -a $O/per_call_period.txt
find $O/pc_trace -name "*.csv" | head; python tools/per_call_period.py --trace $O/pc_trace | tee -a $O/per_call_period.txt
rm -rf $O/pc_trace
timeout 400 bash tools/bench_2proc_sim.sh 2>&1 | tee $O/bench_2proc_sim.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --rccl --no-cpu-baseline --no-per-call --also none > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err; echo "rccl bench rc=$? lines=$(wc -l < $O/bench_rccl_world1.json)"; python -c "
import json; d=json.load(open('$O/bench_rccl_world1.json')); print('rccl_ranks', d['rccl_ranks'], 'return_allgather_ms', d['return_allgather_ms'], 'backend', d['collective_backend'], 'value %.3e' % d['value'])"
