#!/bin/bash
# A/B of library variants for one family (ENV=...) on one box: tools/ab_family.sh VARIANT...  (libs in gpurun_in/libcarl_<V>.so; "base" = product lib)
export CARL_AMD_NO_BUILD=1
for rep in 1 2; do
for v in base "$@"; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  python bench.py --env ${ENV:-cartpole} --steps 300 --warmup 30 --no-cpu-baseline --no-per-call --also none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'launch_us %.2f  ns/step %.1f  value %.3e'%(r['avg_launch_ms']*1e3, r['avg_launch_ms']*1e6/d['config']['chunk'], d['value']))"
done; done
