#!/bin/bash
export CARL_AMD_NO_BUILD=1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_api.py tests/test_gpu_mixed_and_multiproc.py -m gpu -q 2>&1 | tail -3
for e in pendulum mountaincar mountaincar_cont; do ENV=$e bash tools/ab_cartpole.sh PREV | sed "s/^/$e /"; done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
