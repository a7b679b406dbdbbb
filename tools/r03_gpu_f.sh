#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_brax.py tests/test_gpu_brax_invariants.py tests/test_brax_physics_kat.py tests/test_gpu_mixed_and_multiproc.py -q -m gpu -p no:cacheprovider -x > $O/pytest_brax.log 2>&1; tail -3 $O/pytest_brax.log | cut -c1-300
VARIANTS="${VARIANTS:-wg4_w3 wg1_w2 wg2_w3 wg4_w4}" bash tools/r03_gpu_e.sh 2>&1 | grep value | tee $O/brax_variants.txt
