#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_brax.py -q -m gpu -p no:cacheprovider -x > $O/pytest_brax.log 2>&1; tail -3 $O/pytest_brax.log | cut -c1-300
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03h/widths.txt
import sys; sys.argv=[sys.argv[0]]
import torch, bench
dev = torch.device("cuda", 0)
for fam in ("ant", "halfcheetah", "humanoid"):
    wl = bench.Workload((fam,), 32768, 20, 2, 0, 1, dev)
    eng = wl.eng
    print(fam, "autotune ms per 2-step rollout:", {k: round(v, 3) for k, v in eng.autotune_ms.items()}, "-> picked", eng.sys.lanes_per_env)
    wall, avg = wl.train(10, 3, lambda: None)
    print(fam, "value %.3e" % (wl.n * 20 * 10 / wall), "ms/launch %.3f" % (avg * 1e3))
PY
