#!/bin/bash
bash tools/ab_brax_parity.sh BX_F64REL BX_F64RELE
VARIANTS="BX_F64REL BX_F64RELE" ENVS="ant humanoid" bash tools/ablate_brax.sh
for e in ant humanoid; do python bench.py --env $e --lanes 32768 --steps 100 --warmup 20 --no-cpu-baseline --no-per-call --also none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e base %.3e launch_ms %.3f'%(d['value'], d['roofline']['avg_launch_ms']))"; done
