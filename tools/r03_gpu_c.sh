#!/bin/bash
# round 3, GPU call C: full GPU suite (new tests), mass combos, per-call period vs rocprofv3 durations, bench line
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -rs > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log | cut -c1-600
timeout 300 python tools/mass_combo_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/mass_combo_sweep.txt
timeout 200 python tools/per_call_period.py 2>&1 | grep -v amdgpu.ids | tee $O/per_call_period.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/pc_trace -o pc -- python $OLDPWD/tools/per_call_period.py > $OLDPWD/$O/per_call_period_under_rocprof.txt 2>&1) ; grep -v amdgpu.ids $O/per_call_period_under_rocprof.txt | tail -3
python tools/per_call_period.py --trace $O/pc_trace | tee -a $O/per_call_period.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03c/bench_driver_cmd.json"))
print("HEADLINE", d["metric"], "|", d["config"]["workload"][:60], "value %.3e" % d["value"], "ms %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"])
print("sustained", d["sustained"])
print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"], "per_call", {k: v for k, v in (d["per_call"] or {}).items() if k != "roofline"})
for k, v in d["also"].items():
    print(k, "%.3e" % v["value"], "ms %.3f" % v["ms_per_step"], "frac %.3f" % v["frac"], v.get("classes"), v["lanes_per_env"])
PY
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --rccl --no-cpu-baseline --no-per-call --also none > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err; echo "rccl bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_rccl_world1.json')); print('rccl_ranks', d['rccl_ranks'], 'return_allgather_ms', d['return_allgather_ms'], 'backend', d['collective_backend'], 'value %.3e' % d['value'])"
