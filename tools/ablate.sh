#!/bin/bash
# ablation sweep on the GPU box: tools/ablate.sh  (variant libraries prebuilt under gpurun_in/)
export CARL_AMD_NO_BUILD=1
mkdir -p gpurun_out
run() {  # name, extra bench args...
  local name=$1; shift
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-per-call --also none "$@" > gpurun_out/abl_$name.log 2>&1
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
for e in ${ENVS:-pendulum mountaincar}; do
  run base_$e --env $e
  run base_T1000_$e --env $e --chunk 1000 --steps 50 --warmup 5
  for v in NO_DRAIN NO_LOADER NO_BOTH NO_SINK NO_ALL; do
    CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so run ${v}_$e --env $e
  done
done
