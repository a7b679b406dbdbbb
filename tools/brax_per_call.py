"""Per-call carl_brax_step rate (one launch per env step, hipGraph-free eager calls) and short rollouts:
python tools/brax_per_call.py [n_envs]      (on the GPU box; CARL_AMD_LIB_PATH selects the library)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from carl_amd import envs as E  # noqa: E402
from carl_amd.brax_engine import BraxVecEngine  # noqa: E402
from carl_amd.envs.brax.models import SYSTEMS  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for cls_name in ("CARLBraxAnt", "CARLBraxHalfcheetahStiffness", "CARLBraxHumanoidStiffness"):
    cls = getattr(E, cls_name)
    feats = cls.get_context_features()
    names = list(feats)
    rows = np.tile([float(f.default_value) for f in feats.values()], (n, 1)).astype(np.float32).astype(np.float64)
    s = SYSTEMS[cls.env_name](names)
    eng = BraxVecEngine(s, len(names), rows, n, "cuda", selector=0, ctx_idx0=np.arange(n), seed=3, max_episode_steps=1000)
    eng.reset()
    line = f"{cls_name:30s}"
    for T in (1, 2, 5):
        eng.autotune(n_steps=T, reps=2)
        a = torch.rand((T, n, s.n_act), device="cuda") * 2 - 1
        out = eng.alloc_rollout(T)
        for _ in range(5):
            eng.rollout(a, out) if T > 1 else eng.step(a[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 40
        for _ in range(reps):
            eng.rollout(a, out) if T > 1 else eng.step(a[0])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        line += f"  T={T}: {dt * 1e6:7.1f} us/launch {n * T / dt:.3e} env-steps/s (width {eng.sys.lanes_per_env})"
    print(line, flush=True)
