#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
for e in ant humanoid halfcheetah; do
  tools/pmc_sq.sh r03k_$e --env $e --lanes 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-per-call --also none > $O/${e}_sq_counters.txt 2>&1
done
cat $O/ant_sq_counters.txt | head -5
rm -rf gpurun_out/pmc_r03k_*
