#!/bin/bash
export CARL_AMD_NO_BUILD=1
mkdir -p gpurun_out
run() {
  local name=$1; shift
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-per-call "$@" > gpurun_out/abl_$name.log 2>&1
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
txt = open(f"gpurun_out/abl_{n}.log").read()
l = [x for x in txt.splitlines() if x.startswith("{")]
if not l: print(n, "FAILED", txt[-400:])
else:
    d = json.loads(l[-1]); r = d["roofline"]
    print(f"{n:28s} launch_ms {r['avg_launch_ms']:.4f}  ns/step {r['avg_launch_ms']*1e6/d['config']['chunk']:.1f}")
PY
}
for rep in $(seq 1 ${REPS:-2}); do
for v in ${VARIANTS}; do
  CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so run ${v}_r$rep --env ${ENV:-pendulum} $ARGS
done
done
