#!/bin/bash
# LDS counters of the Brax rollout kernel: tools/pmc_lds.sh <tag> [bench args]   (own rocprofv3 passes, --pmc only)
set -u
TAG=${1:-lds}; shift || true
ARGS=${*:-"--env ant --lanes 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-per-call --also none"}
export TMPDIR=/tmp CARL_AMD_NO_BUILD=1
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_WAVE_CYCLES -d "$OUT/a" -o bench --output-format csv -- python bench.py $ARGS > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU -d "$OUT/b" -o bench --output-format csv -- python bench.py $ARGS > "$OUT/b.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "brax_kernel<1" not in k:
                continue
            a = acc[k[:70]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    if not acc:
        print(sub, "no rows:", open(f"{root}/{sub}.log").read()[-600:])
    for k, d in acc.items():
        print(k)
        for c, (n, t) in sorted(d.items()):
            print(f"   {c:24s} avg/dispatch {t / n:16.1f}  dispatches {n}")
PY
