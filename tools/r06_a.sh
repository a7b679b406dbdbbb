export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'frac',d['roofline']['frac'],d['roofline'].get('frac_launch_median'))
for k,v in d['also'].items(): print(k, '%.3e'%v['value'], (v.get('cpu_baseline') or {}).get('value'))
print('cpu', d['cpu_baseline'])
PY
