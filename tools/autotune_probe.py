"""What the lane-group width autotune sees at two probe lengths (2 env steps: BraxVecEngine.autotune's default; 20: the
bench's launch length): python tools/autotune_probe.py [n_envs]      (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from carl_amd import envs as E  # noqa: E402
from carl_amd.brax_engine import BraxVecEngine  # noqa: E402
from carl_amd.envs.brax.models import SYSTEMS  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for cls_name in ("CARLBraxAnt", "CARLBraxHalfcheetahStiffness", "CARLBraxHumanoidStiffness", "CARLBraxHopper", "CARLBraxWalker2d"):
    cls = getattr(E, cls_name)
    feats = cls.get_context_features()
    names = list(feats)
    rows = np.tile([float(f.default_value) for f in feats.values()], (n, 1)).astype(np.float32).astype(np.float64)
    s = SYSTEMS[cls.env_name](names)
    eng = BraxVecEngine(s, len(names), rows, n, "cuda", selector=0, ctx_idx0=np.arange(n), seed=3, max_episode_steps=1000)
    eng.reset()
    for T in (2, 20):
        best = eng.autotune(n_steps=T, reps=3)
        print(f"{cls_name:30s} T={T:2d} best {best:2d}  ms per launch " + "  ".join(f"{w}: {ms:.3f}" for w, ms in eng.autotune_ms.items()), flush=True)
