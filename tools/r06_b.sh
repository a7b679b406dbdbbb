#!/bin/bash
# Round 6, second GPU call: the suite on the new gear tables, soak of every Brax family, the 8-rank self-launch on ONE GPU
# (plumbing only), Brax parity records on the new tables, and the costing of experiment E_A (float32 anchor rotations in the
# lean kernel: gpurun_in/libcarl_f32anchor.so) -- time A/B and its parity.
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r06b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python tools/soak_brax.py 8192 1000 > $O/soak_brax_all_families.txt 2>&1; tail -12 $O/soak_brax_all_families.txt
CARL_BENCH_SHARE_GPU=1 CARL_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_selflaunch_8ranks.json 2> $O/bench_selflaunch_8ranks.err; echo "8rank rc $?"; tail -2 $O/bench_selflaunch_8ranks.err
timeout 900 python tools/brax_parity_long.py 16384 100 ant halfcheetah humanoid > $O/brax_parity_long_base.txt 2>&1; tail -4 $O/brax_parity_long_base.txt
CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_f32anchor.so timeout 900 python tools/brax_parity_long.py 16384 100 ant halfcheetah > $O/brax_parity_long_f32anchor.txt 2>&1; tail -3 $O/brax_parity_long_f32anchor.txt
ENVS="ant halfcheetah" NO_DIGEST=1 bash tools/ab_brax.sh f32anchor > $O/ab_f32anchor.txt 2>&1; tail -10 $O/ab_f32anchor.txt
