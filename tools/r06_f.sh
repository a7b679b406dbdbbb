#!/bin/bash
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r06f; mkdir -p $O
for v in base limit64; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  timeout 900 python tools/brax_parity_long.py 16384 300 humanoid 2>&1 | grep -v amdgpu > $O/parity_long_$v.txt; cat $O/parity_long_$v.txt
  timeout 300 python tools/brax_parity_percentiles.py humanoid humanoidstandup 2>&1 | grep -v amdgpu > $O/pct_$v.txt; cat $O/pct_$v.txt
done
unset CARL_AMD_LIB_PATH
ENVS="humanoid" bash tools/ab_brax.sh limit64 2>&1 | tail -5
