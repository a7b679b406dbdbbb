#!/bin/bash
# Round-4 closing evidence (one gpurun call): the driver's command plain and under rocprofv3 --kernel-trace --stats,
# workgroup placement, the 2-process plumbing run.
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r04_final; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/kt.log 2>&1
python tools/summarize_profile.py $O > $O/driver_cmd_rocprofv3.txt 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_under_rocprofv3.json
tools/wg_placement/wg_placement 512 512 77824 > $O/wg_placement.txt 2>&1
bash tools/bench_2proc_sim.sh > $O/bench_2proc_sim.txt 2>&1
head -40 $O/driver_cmd_rocprofv3.txt; tail -5 $O/wg_placement.txt; cat $O/bench_2proc_sim.txt | tail -9
python tools/trace_by_shape.py $O > $O/driver_cmd_trace_by_shape.txt 2>&1; cat $O/driver_cmd_trace_by_shape.txt
