#!/bin/bash
# Round-6 closing evidence (one gpurun call).  VERDICT r05 "Next" #1: every committed figure of the headline from ONE call
# on ONE box, on the round's FINAL binary:
#   (1) the driver's command plain -> under rocprofv3 --kernel-trace --stats -> plain again (the pair whose trace average
#       <= the plain line's ms_per_step is the one to cite; all three are committed);
#   (2) tools/r04_evidence.sh r06: focused rocprofv3 passes per workload (kernel trace; FETCH / WRITE / SQ counters in
#       separate passes) -> afterwards, locally:  python tools/make_r04_profiles.py gpurun_out/prof_r06 "<label>" r06
#       writes profiles/kernel_times.json, traffic.json, brax_valu.json AND profiles/r06_rocprofv3_summary.txt with ONE
#       source label (tests/test_bench_contract.py checks that they agree);
#   (3) Brax parity records and the soak on the final binary.
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
O=$PWD/gpurun_out/r06_final; mkdir -p $O
python - > $O/binary.txt <<'PY'
from carl_amd import build as b
print("source_hash", b._source_hash()[:16], "needs_build", b.needs_build())
PY
cat $O/binary.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_1.json 2> $O/bench_driver_cmd_1.err
# (under the profiler WITHOUT the 256-process CPU baseline: every spawned worker loads the rocprofv3 tool, and its signal
#  handlers can deadlock the pool's teardown -- r05_final.sh)
rm -rf $O/kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1
python tools/summarize_profile.py $O > $O/driver_cmd_rocprofv3.txt 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_under_rocprofv3.json
python tools/trace_by_shape.py $O > $O/driver_cmd_trace_by_shape.txt 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_2.json 2> $O/bench_driver_cmd_2.err
rm -rf $O/kt/*/*.json  # (keep the CSVs; the merged scratch is capped at 64 MiB)
python - <<'PY'
import json
for f in ("bench_driver_cmd_1", "bench_under_rocprofv3", "bench_driver_cmd_2"):
    try:
        d = json.loads(open(f"gpurun_out/r06_final/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "value %.4e ms_per_step %.4f frac %.3f launch_median %.3f" % (d["value"], d["ms_per_step"], r["frac"], r.get("frac_launch_median") or 0),
              "config4 %.3e config5 %.3e" % (d["also"]["config4"]["value"], d["also"]["config5"]["value"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -12 $O/driver_cmd_rocprofv3.txt
bash tools/r04_evidence.sh r06 > $O/evidence.log 2>&1; tail -3 $O/evidence.log
find gpurun_out/prof_r06 -name "*.json" -size +1M -delete; find gpurun_out/prof_r06 -name "*agent_info.csv" -delete
timeout 600 python tools/brax_parity_percentiles.py > $O/brax_parity_percentiles.txt 2>&1
timeout 1500 python tools/brax_parity_long.py 16384 300 ant halfcheetah humanoid hopper walker2d > $O/brax_parity_long.txt 2>&1; tail -3 $O/brax_parity_long.txt
timeout 900 python tools/soak_brax.py 32768 1000 > $O/soak_brax_all_families.txt 2>&1; tail -3 $O/soak_brax_all_families.txt
du -sh gpurun_out
