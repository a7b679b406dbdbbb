#!/bin/bash
# Round-4 profiler evidence (run on the GPU box via gpurun): per BASELINE workload ONE focused bench invocation under
#   rocprofv3 --kernel-trace --stats      -> the kernel's own average duration (profiles/kernel_times.json: frac_kernel)
#   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, never with a tracing domain) -> HBM traffic per launch
#   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES (Brax workloads)     -> the VALU-issue roofline
# Focused = that workload alone, full size, launch shape pinned (--lanes-per-env): every dispatch of the rollout
# kernel in the trace is one of the timed full-size launches -- no autotune probes, no half-batch launches.
#   tools/r04_evidence.sh [tag]      then   python tools/make_r04_profiles.py gpurun_out/prof_<tag> "<source label>"
set -u
TAG=${1:-r04}
ONLY=${2:-}   # optional: a grep pattern over the workload names (e.g. narrow)
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
COMMON="--steps 20 --warmup 5 --reps 3 --also none --no-shard8 --no-cpu-baseline --no-per-call --sustained-seconds 0.1"
run() {  # name, bench args...
  local name=$1; shift
  if [ -n "$ONLY" ] && ! echo "$name" | grep -q "$ONLY"; then return; fi
  for pass in kt fetch write sq; do
    case $pass in
      kt) flags="--kernel-trace --stats" ;;
      fetch) flags="--pmc FETCH_SIZE" ;;
      write) flags="--pmc WRITE_SIZE" ;;
      sq) case $name in ant*|halfcheetah*) flags="--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" ;; *) continue ;; esac ;;
    esac
    timeout 240 rocprofv3 $flags -d "$OUT/$name/$pass" -o bench --output-format csv -- python bench.py $COMMON "$@" > "$OUT/$name.$pass.log" 2>&1
    tail -1 "$OUT/$name.$pass.log" | cut -c1-200
  done
}
run cartpole_65536_1000 --env cartpole
run pendulum_65536_1000 --env pendulum
run acrobot+mountaincar_65536_1000 --env acrobot+mountaincar
run cartpole_65536_250 --env cartpole --chunk 250
run pendulum_65536_250 --env pendulum --chunk 250
run acrobot+mountaincar_65536_250 --env acrobot+mountaincar --chunk 250
run ant_32768_20 --env ant --lanes 32768 --lanes-per-env ant=9
run halfcheetah+humanoid_32768_20 --env halfcheetah+humanoid --lanes 32768 --lanes-per-env halfcheetah=7,humanoid=11
run cartpole_8192_1000 --env cartpole --lanes 8192
run cartpole_narrow_65536_1000 --env cartpole --narrow-actions
run pendulum_narrow_65536_1000 --env pendulum --narrow-actions
run ant_4096_20 --env ant --lanes 4096 --lanes-per-env ant=16
find "$OUT" -name "*kernel_stats.csv" | head -20
