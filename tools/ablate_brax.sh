#!/bin/bash
# brax kernel variants: VARIANTS="W2 W3" ENVS="ant humanoid" tools/ablate_brax.sh
export CARL_AMD_NO_BUILD=1
mkdir -p gpurun_out
for e in ${ENVS:-ant humanoid halfcheetah}; do
for v in ${VARIANTS}; do
  CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so timeout 200 python bench.py --env $e --lanes ${LANES:-32768} --chunk 20 --steps 100 --warmup 20 --no-cpu-baseline --no-per-call --also none > gpurun_out/bb_${e}_$v.log 2>&1
  python - $e $v <<'PY'
import json,sys
e,v=sys.argv[1:3]
l=[x for x in open(f"gpurun_out/bb_{e}_{v}.log") if x.startswith("{")]
if not l: print(e, v, "FAILED", open(f"gpurun_out/bb_{e}_{v}.log").read()[-300:])
else:
    d=json.loads(l[-1]); print(f"{e:12s} {v:8s} {d['value']:.3e} env-steps/s  launch_ms {d['roofline']['avg_launch_ms']:.3f}")
PY
done
done
