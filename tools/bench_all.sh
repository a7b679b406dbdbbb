#!/bin/bash
# quick per-family bench summary on the GPU box: tools/bench_all.sh [extra bench.py args]
export CARL_AMD_NO_BUILD=1
mkdir -p gpurun_out
for e in ${ENVS:-pendulum cartpole acrobot mountaincar}; do
  python bench.py --env $e --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --also none "$@" > gpurun_out/bench_$e.log 2>&1
  python - "$e" <<'PY'
import json, sys
e = sys.argv[1]
txt = open(f"gpurun_out/bench_{e}.log").read()
l = [x for x in txt.splitlines() if x.startswith("{")]
if not l:
    print(e, "FAILED:", txt[-600:])
else:
    d = json.loads(l[-1]); r = d["roofline"]; p = d["per_call"]
    s = f"{e:12s} value {d['value']:.3e} launch_ms {r['avg_launch_ms']:.4f} frac {r['frac']:.3f} (8d-bytes {r['achieved_with_survey_8d_bytes']/8000:.3f})"
    if p and p.get("graph_value"):
        s += f" | per-call eager {p['eager_value']:.2e} graph {p['graph_value']:.2e} ({p['graph_ms_per_step']*1e3:.2f} us/step)"
    print(s)
PY
done
