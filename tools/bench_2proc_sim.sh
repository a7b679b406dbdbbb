#!/bin/bash
# Exercise bench.py's N > 1 code path on a ONE-GPU box: two ranks share GPU 0 and the collectives run over
# gloo (RCCL refuses two ranks on one device).  Checks the rank plumbing, lane shards, barriers, max-over-ranks
# timing, the strong-scaling `also` records and the episodic-return all-gather -- not performance (the real
# 2/4/8-GPU runs are the driver's).
export CARL_AMD_NO_BUILD=1 CARL_BENCH_SHARE_GPU=1 CARL_BENCH_BACKEND=gloo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 10 --warmup 3 2>&1 | grep "^{" | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n_gpus', d['n_gpus'], 'value %.3e' % d['value'], 'total_lanes', d['config']['total_lanes'], 'rccl_ranks', d['rccl_ranks'],
      'allgather_ms', d['return_allgather_ms'], 'per_rank_launch_ms', d['per_rank_avg_launch_ms'], 'backend', d['collective_backend'])
o = d.get('strong') or d.get('weak')
print('scaling', d['scaling'], 'lanes/gpu', d['config']['lanes_per_gpu'], '| other record:', o and {k: o[k] for k in ('scaling', 'value', 'lanes_per_gpu', 'total_lanes')})
for k, v in d['also'].items(): print('  ', k, '%.3e' % v['value'], v['scaling'], 'lanes/gpu', v['lanes_per_gpu'])
"
