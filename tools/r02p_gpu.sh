#!/bin/bash
export CARL_AMD_NO_BUILD=1
timeout 300 python -m pytest tests/test_gpu_brax.py tests/test_gpu_brax_invariants.py -m gpu -q 2>&1 | tail -3
for e in ant humanoid halfcheetah; do
 for v in base PREV; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  timeout 60 python bench.py --env $e --lanes 32768 --steps 60 --warmup 10 --no-cpu-baseline --no-per-call --also none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e $v %.3e launch_ms %.3f'%(d['value'], d['roofline']['avg_launch_ms']), d['config']['lanes_per_env'])"
 done
done
unset CARL_AMD_LIB_PATH
python tools/brax_parity_percentiles.py ant humanoid halfcheetah 2>&1 | grep -v amdgpu.ids
