#!/bin/bash
# Round 6, fourth GPU call: CARL_FLAG_BRAX_FP32 -- tests, deviation from the float64 restatement, rate beside the product path
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_brax.py tests/test_gpu_brax_invariants.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
CARL_BRAX_FP32=1 timeout 600 python tools/brax_parity_percentiles.py ant halfcheetah humanoid hopper walker2d inverted_pendulum inverted_double_pendulum humanoidstandup 2>&1 | grep -v amdgpu.ids > $O/brax_fp32_deviation.txt; cat $O/brax_fp32_deviation.txt
timeout 300 python tools/brax_parity_percentiles.py ant halfcheetah humanoid 2>&1 | grep -v amdgpu.ids > $O/brax_fp64_percentiles.txt; cat $O/brax_fp64_percentiles.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-per-call --no-shard8 --also config4,config5,config4_fp32,config5_fp32 > $O/bench_brax.json 2> $O/bench_brax.err; tail -3 $O/bench_brax.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06d/bench_brax.json').read().strip().splitlines()[-1])
for k,v in d['also'].items(): print(k, '%.3e'%v['value'], 'launch ms %.3f'%v['avg_launch_ms'], v.get('lanes_per_env'), v.get('deviation'))
PY
