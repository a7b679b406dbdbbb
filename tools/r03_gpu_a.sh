#!/bin/bash
# round 3, GPU call A: fp64 issue rate, Brax parity (branch-aware), Brax GPU tests, Brax throughput
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout 120 tools/fp64_rate/fp64_rate > $O/fp64_rate.txt 2>&1; tail -16 $O/fp64_rate.txt
timeout 420 python tools/brax_parity_percentiles.py 2>&1 | grep -v amdgpu.ids > $O/brax_parity.txt; cat $O/brax_parity.txt
timeout 600 python -m pytest tests/test_gpu_brax.py tests/test_gpu_brax_invariants.py -q -m gpu -p no:cacheprovider -s > $O/pytest_brax.log 2>&1; tail -40 $O/pytest_brax.log | cut -c1-400
for e in ant halfcheetah humanoid; do
  timeout 200 python bench.py --env $e --lanes 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-per-call --also none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['workload'][:40], 'value %.3e' % d['value'], 'ms/launch %.3f' % d['ms_per_step'], d['config']['lanes_per_env'])"
done 2>&1 | tee $O/bench_brax.txt
