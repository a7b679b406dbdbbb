#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
run() { # name libpath env sub
  CARL_AMD_BRAX_SUB=$4 CARL_AMD_LIB_PATH=$2 timeout 200 python bench.py --env $3 --lanes 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-per-call --also none 2>/tmp/bench_err.txt > /tmp/bench_out.txt
  python - "$1" "$3" "$4" <<'PY'
import sys, json
t = open('/tmp/bench_out.txt').read()
if not t.strip():
    print(sys.argv[1:], 'FAILED:', open('/tmp/bench_err.txt').read()[-600:].replace(chr(10), ' | '))
else:
    d = json.loads(t.strip().splitlines()[-1]); print(sys.argv[1:], 'value %.3e' % d['value'], 'ms/launch %.3f' % d['ms_per_step'])
PY
}
for v in product w3 w4; do
  lib=""; [ $v != product ] && lib=$PWD/gpurun_in/libcarl_$v.so
  run $v "$lib" ant 9; run $v "$lib" halfcheetah 7; run $v "$lib" humanoid 11
done 2>&1 | tee $O/occupancy.txt
