#!/bin/bash
# round 3, GPU call B: full GPU suite + SQ counters of the Brax kernel (Ant, Humanoid)
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log | cut -c1-300
for e in ant humanoid; do
  tools/pmc_sq.sh r03_$e --env $e --lanes 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-per-call --also none > $O/${e}_sq_counters.txt 2>&1
  cat $O/${e}_sq_counters.txt
done
