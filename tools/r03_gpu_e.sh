#!/bin/bash
# round 3, GPU call E: Brax kernel occupancy variants and cost attribution (measurement-only libraries in gpurun_in/)
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
run() { # name libpath
  for e in ant halfcheetah humanoid; do
    CARL_AMD_LIB_PATH=$2 timeout 200 python bench.py --env $e --lanes 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-per-call --also none 2>/tmp/bench_err.txt | python -c "
import sys, json
t = sys.stdin.read()
if not t.strip(): print('$1', '$e', 'FAILED:', open('/tmp/bench_err.txt').read()[-400:].replace(chr(10), ' | ')); sys.exit(0)
d = json.loads(t); print('$1', '$e', 'value %.3e' % d['value'], 'ms/launch %.3f' % d['ms_per_step'], d['config']['lanes_per_env'])"
  done
}
{ run product ""; for v in ${VARIANTS:-w3 w4 fastatan nocontacts noA noB}; do run $v $PWD/gpurun_in/libcarl_$v.so; done; } 2>&1 | tee $O/brax_variants.txt
