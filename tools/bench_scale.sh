#!/bin/bash
# The exact lines of the N = 1 / 2 / 4 / 8 scaling run on one 8-GPU MI355X node (what the driver launches; VERDICT r02
# #7).  One process per GPU over RCCL; each line prints ONE JSON record (rank 0): `value` = weak scaling (65 536 contexts
# per GPU, whole-node aggregate), `strong` = BASELINE's 65 536 contexts split over the node, `also.config4 / config5`
# split BASELINE's Brax totals over the ranks.
#   tools/bench_scale.sh [steps] [warmup]  ->  gpurun_out/scale/bench_N<k>.json
K=${1:-20}; W=${2:-5}
O=gpurun_out/scale; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps $K --warmup $W > $O/bench_N1.json
for N in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
    bench.py --gpus $N --steps $K --warmup $W > $O/bench_N$N.json
done
python - <<'PY'
import json
base = None
for n in (1, 2, 4, 8):
    try:
        d = json.loads(open(f"gpurun_out/scale/bench_N{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "missing", e)
        continue
    base = base or d["value"]
    st = d.get("strong") or {"value": d["value"]}
    print(f"N={n}: weak {d['value']:.3e} ({d['value'] / base:.2f}x of N=1)   strong {st['value']:.3e} ({st['value'] / base:.2f}x)   "
          f"allgather {d['return_allgather_ms']} ms over {d['rccl_ranks']} ranks")
PY
