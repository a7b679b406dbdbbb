#!/bin/bash
# A/B: the float32-substep kernels at 3 / 4 wavefronts per SIMD (multi / lean) against 2 / 3 (gpurun_in/libcarl_f32occ.so)
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r06e; mkdir -p $O
for rep in 1 2; do for v in base f32occ; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-per-call --no-shard8 --also config4_fp32,config5_fp32 --env pendulum --lanes 4096 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['also'].items(): print('$v', k, '%.3e'%v['value'], 'launch ms %.3f'%v['avg_launch_ms'], v.get('lanes_per_env'))"
done; done 2>&1 | tee $O/ab_f32occ.txt
