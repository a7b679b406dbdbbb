#!/bin/bash
# After `gpurun -- bash tools/r06_final.sh` has merged gpurun_out/: write the tracked records under profiles/ (one source
# label for kernel_times / traffic / brax_valu / the summary, naming the binary's source hash the call printed).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r06_final
H=$(awk '/source_hash/ {print $2}' $O/binary.txt)
python tools/make_r04_profiles.py gpurun_out/prof_r06 "tools/r04_evidence.sh r06 (rocprofv3, one focused bench invocation per workload, launch shape pinned), round-6 final binary, source hash $H (carl_amd.build), same gpurun call as profiles/r06_bench_driver_cmd_*.json" r06 > /dev/null
for f in bench_driver_cmd_1.json bench_driver_cmd_2.json bench_under_rocprofv3.json driver_cmd_rocprofv3.txt driver_cmd_trace_by_shape.txt; do cp $O/$f profiles/r06_$f; done
for f in brax_parity_percentiles brax_parity_long; do grep -v amdgpu $O/$f.txt > profiles/r06_$f.txt; done
{ grep -v amdgpu $O/soak_brax_all_families.txt; if [ -f gpurun_out/soak_config5_contexts.txt ]; then echo; echo "== BASELINE config 5's context distribution (joint_stiffness x U(0.5, 2), gravity U(-15, -5)): tools/soak_config5_contexts.py"; cat gpurun_out/soak_config5_contexts.txt; fi; } > profiles/r06_soak_brax_all_families.txt
echo "published: source hash $H"; python -c "
from carl_amd import build as b
print('tree hash', b._source_hash()[:16], 'needs_build', b.needs_build())"
grep -n 'CartPole' profiles/r06_driver_cmd_trace_by_shape.txt | head -3
