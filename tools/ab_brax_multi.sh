#!/bin/bash
# Launch times of several builds of the library on ONE box, interleaved: tools/ab_brax_multi.sh "ant humanoid" base v1 v2 ...
# ("base" = the product library; other names = gpurun_in/libcarl_<name>.so from tools/build_variant.sh)
export CARL_AMD_NO_BUILD=1
envs=$1; shift
for e in $envs; do for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  python bench.py --env $e --lanes ${LANES:-32768} --steps 40 --warmup 5 --no-cpu-baseline --no-per-call --also none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %-8s launch_ms %.3f value %.3e'%('$e','$v', r['avg_launch_ms'], d['value']))"
done; done; done
