#!/bin/bash
# Brax throughput of the product library and of measurement-only variants built by tools/build_variant.sh (gpurun_in/):
#   VARIANTS="product w4" [TEST_LIB=w4] tools/brax_variants.sh      (on the GPU box; profiles/r03_brax_occupancy.txt was made with it)
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/brax_variants; mkdir -p $O
run() { # name libpath env
  CARL_AMD_LIB_PATH=$2 timeout 120 python bench.py --env $3 --lanes 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-per-call --also none 2>/tmp/bench_err.txt > /tmp/bench_out.txt
  python - "$1" "$3" <<'PY'
import sys, json
t = open('/tmp/bench_out.txt').read()
if not t.strip():
    print(sys.argv[1:], 'FAILED:', open('/tmp/bench_err.txt').read()[-400:].replace(chr(10), ' | '))
else:
    d = json.loads(t.strip().splitlines()[-1]); print(sys.argv[1:], 'value %.3e' % d['value'], 'ms/launch %.3f' % d['ms_per_step'], d['config']['lanes_per_env'])
PY
}
for v in ${VARIANTS:-product w2_inl w3_inl}; do
  lib=""; [ $v != product ] && lib=$PWD/gpurun_in/libcarl_$v.so
  run $v "$lib" ant; run $v "$lib" halfcheetah; run $v "$lib" humanoid
done 2>&1 | tee $O/variants.txt
if [ -n "$TEST_LIB" ]; then
  CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$TEST_LIB.so timeout 900 python -m pytest tests/test_gpu_brax.py tests/test_gpu_brax_invariants.py tests/test_brax_physics_kat.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3 | cut -c1-300
fi
