#!/bin/bash
# Round 6, third GPU call: the suite on the round-6 Humanoid tables (spring-branch gears + humanoid.xml's constraint
# constants), stability of every env (soak; HumanoidStandup sweep), re-measured mass floors for the families whose tables changed.
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python tools/diag_standup_sweep.py 16384 1000 2>&1 | grep -v amdgpu.ids > $O/standup_gear_stability.txt; cat $O/standup_gear_stability.txt
timeout 600 python tools/mass_stability_sweep.py humanoid humanoidstandup halfcheetah 2>&1 | grep -v amdgpu.ids > $O/mass_stability_sweep.txt; cat $O/mass_stability_sweep.txt
timeout 600 python tools/mass_combo_sweep.py humanoid humanoidstandup halfcheetah 2>&1 | grep -v amdgpu.ids > $O/mass_combo_sweep.txt; cat $O/mass_combo_sweep.txt
timeout 900 python tools/soak_brax.py 32768 1000 2>&1 | grep -v amdgpu.ids > $O/soak_brax_all_families.txt; cat $O/soak_brax_all_families.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-per-call --no-shard8 --also config4,config5,config4_T100,config5_T100 > $O/bench_brax.json 2> $O/bench_brax.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06c/bench_brax.json').read().strip().splitlines()[-1])
for k,v in d['also'].items(): print(k, '%.3e'%v['value'], 'launch ms %.3f'%v['avg_launch_ms'], v.get('lanes_per_env'))
PY
