#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | cut -c1-300; grep -n "^____\|^E  " $O/pytest_gpu.log | head -20
timeout 400 bash tools/bench_2proc_sim.sh 2>&1 | tee $O/bench_2proc_sim.txt
