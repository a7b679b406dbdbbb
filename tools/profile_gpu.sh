#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box via gpurun):
#   tools/profile_gpu.sh <tag> [bench.py args...]
# 1. --kernel-trace --stats  -> per-kernel average duration (must agree with bench.py's
#    live HIP-event figure)
# 2./3. --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (TCC slots: 3 + 2 > 4;
#    never combined with tracing domains) -> HBM traffic per launch.
set -u
TAG=${1:-r01}; shift || true
ARGS=${*:-"--steps 200 --warmup 20 --no-cpu-baseline --no-per-call --also none"}
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o bench --output-format csv -- python bench.py $ARGS > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench --output-format csv -- python bench.py $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench --output-format csv -- python bench.py $ARGS > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
python tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
