#!/bin/bash
# SQ counters for the rollout kernel: tools/pmc_sq.sh <tag> [bench args]
set -u
TAG=${1:-sq}; shift || true
ARGS=${*:-"--steps 200 --warmup 20 --no-cpu-baseline --no-per-call --also none"}
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/a" -o bench --output-format csv -- python bench.py $ARGS > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU -d "$OUT/b" -o bench --output-format csv -- python bench.py $ARGS > "$OUT/b.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rollout" not in k and "step_kernel" not in k and "brax_kernel<1" not in k:
                continue
            a = acc[k[:70]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, d in acc.items():
        print(k)
        for c, (n, t) in sorted(d.items()):
            print(f"   {c:24s} avg/dispatch {t / n:16.1f}")
PY
