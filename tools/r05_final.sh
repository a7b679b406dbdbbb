#!/bin/bash
# Round-5 closing evidence (one gpurun call): the driver's command under rocprofv3 --kernel-trace --stats, then plain,
# workgroup placement, the 2-process plumbing run.
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r05_final; mkdir -p $O
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
# round 5: the self-launching N > 1 path on this one GPU, Brax parity records of the FINAL binary, region clocks
CARL_BENCH_SHARE_GPU=1 CARL_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --also none --no-shard8 --no-per-call --no-cpu-baseline > $O/bench_selflaunch_2ranks.json 2> $O/bench_selflaunch_2ranks.err
timeout 600 python tools/brax_parity_percentiles.py > $O/brax_parity_percentiles.txt 2>&1
timeout 1500 python tools/brax_parity_long.py 16384 300 ant halfcheetah humanoid hopper walker2d > $O/brax_parity_long.txt 2>&1
if [ -f gpurun_in/libcarl_prof.so ]; then CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_prof.so timeout 120 python tools/brax_region_profile.py ant halfcheetah humanoid > $O/brax_region_profile.txt 2>&1; fi
tail -3 $O/brax_parity_long.txt
