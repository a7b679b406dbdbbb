"""Stress of the fragment schedule: a batch larger than the chip holds at once, many 20-step launches with auto-reset, two
engines with DIFFERENT lane-group widths (different group sizes, different fragment cuts): every output of every launch
and the final state must be bit-identical.    python tools/stress_fragments.py [family] [n_envs] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from carl_amd import envs as E  # noqa: E402
from carl_amd.brax_engine import BraxVecEngine  # noqa: E402
from carl_amd.envs.brax.models import SYSTEMS  # noqa: E402

fam = sys.argv[1] if len(sys.argv) > 1 else "CARLBraxAnt"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cls = getattr(E, fam)
feats = cls.get_context_features()
names = list(feats)
rng = np.random.default_rng(1)
rows = np.tile([float(f.default_value) for f in feats.values()], (64, 1))
rows[:, names.index("gravity")] = rng.uniform(-15, -5, 64)
rows = rows.astype(np.float32).astype(np.float64)
engs = []
for w in (0, 16):
    s = SYSTEMS[cls.env_name](names)
    if w:
        s.lanes_per_env = w
    e = BraxVecEngine(s, len(names), rows, n, "cuda", selector=1, seed=3, max_episode_steps=37)
    e.reset()
    engs.append(e)
T = 20
amp = float(max(engs[0].sys.act_hi[: engs[0].sys.n_act]))
outs = [e.alloc_rollout(T, final_obs=True) for e in engs]
for k in range(L):
    a = torch.as_tensor(rng.uniform(-amp, amp, (T, n, engs[0].sys.n_act)).astype(np.float32), device="cuda")
    r = [e.rollout(a, o) for e, o in zip(engs, outs)]
    for name in ("obs", "reward", "terminated", "truncated", "final_obs"):
        assert torch.equal(r[0][name], r[1][name]), (k, name)
for name in ("state", "elapsed", "ctx_idx", "episode", "n_calls", "ep_return", "episodes_done", "last_return", "last_length"):
    assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), name
print(f"{fam}: {n} envs x {L} launches x {T} steps, widths {engs[0].sys.lanes_per_env} / {engs[1].sys.lanes_per_env}: every output and the final "
      f"state bit-identical; {int(engs[0].episodes_done.sum())} episodes finished inside the launches")
