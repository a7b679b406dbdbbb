#!/bin/bash
# launch times of library variants on one box: ENVS="ant humanoid" tools/ab_variants.sh V1 V2 ...   ("base" = the product library)
export CARL_AMD_NO_BUILD=1
for e in ${ENVS:-ant halfcheetah humanoid}; do
for rep in 1 2; do for v in base "$@"; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  python bench.py --env $e --lanes ${LANES:-32768} --steps 40 --warmup 5 --no-cpu-baseline --no-per-call --also none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$e $v', 'launch_ms %.3f value %.3e'%(r['avg_launch_ms'], d['value']))"
done; done; done
