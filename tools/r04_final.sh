#!/bin/bash
# Round-4 closing evidence (one gpurun call): the driver's command under rocprofv3 --kernel-trace --stats, then plain,
# workgroup placement, the 2-process plumbing run.
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r04_final; mkdir -p $O
# (under the profiler WITHOUT the 256-process CPU baseline: every spawned worker loads the rocprofv3 tool, and its
#  signal handlers can deadlock the pool's teardown -- one such run hung for the whole 40-minute limit of a gpurun call)
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1
python tools/summarize_profile.py $O > $O/driver_cmd_rocprofv3.txt 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_under_rocprofv3.json
# the kernel's own average durations of THIS box first (profiles/kernel_times.json on the box's copy of the tree), then the
# plain run of the driver's command, whose frac_kernel then comes from the same box and the same call
cp profiles/kernel_times.json $O/kernel_times.before.json
python tools/kernel_times_from_driver_trace.py $O > $O/kernel_times.txt 2>&1; cp profiles/kernel_times.json $O/kernel_times.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 60 tools/wg_placement/wg_placement 512 512 77824 > $O/wg_placement.txt 2>&1
timeout 300 bash tools/bench_2proc_sim.sh > $O/bench_2proc_sim.txt 2>&1
head -40 $O/driver_cmd_rocprofv3.txt; tail -5 $O/wg_placement.txt; cat $O/bench_2proc_sim.txt | tail -9
python tools/trace_by_shape.py $O > $O/driver_cmd_trace_by_shape.txt 2>&1; cat $O/driver_cmd_trace_by_shape.txt
