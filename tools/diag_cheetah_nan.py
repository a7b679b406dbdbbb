import sys; sys.path.insert(0, ".")
import numpy as np, torch
from carl_amd import envs as E
from carl_amd.brax_engine import BraxVecEngine
from carl_amd.envs.brax.models import SYSTEMS
from oracle import brax as B, oracle as O
cls = E.CARLBraxHalfcheetahStiffness
feats = cls.get_context_features(); names = list(feats)
default = np.array([float(f.default_value) for f in feats.values()])
s = SYSTEMS[cls.env_name](names)
n = 4096
rng = np.random.default_rng(1)
rows = np.tile(default, (n, 1))
FEAT = sys.argv[1] if len(sys.argv) > 1 else "joint_stiffness"
RANGE = {"joint_stiffness": (0.5, 2.0), "gravity": (-15, -5), "friction": (0.3, 1.5), "mass_torso": (5, 15)}[FEAT]
rows[:, names.index(FEAT)] = np.linspace(*RANGE, n)
rows = rows.astype(np.float32).astype(np.float64)
kw = dict(selector=O.SEL_STATIC, seed=5, ctx_idx0=np.arange(n))
for scale in (1.0, 1.2):
    eng = BraxVecEngine(s, len(names), rows, n, "cuda", max_episode_steps=10_000, auto_reset=False, **kw)
    eng.reset()
    first_bad = np.full(n, -1)
    for t in range(200):
        a = rng.uniform(-scale, scale, (n, s.n_act)).astype(np.float32)
        obs, rew, term, trunc = eng.step(torch.as_tensor(a))
        bad = (~torch.isfinite(obs).all(1) | (obs.abs().max(1).values > 1e3)).cpu().numpy()
        first_bad[(first_bad < 0) & bad] = t
    k = rows[:, names.index(FEAT)]
    print(FEAT, "action scale", scale, "bad envs", (first_bad >= 0).sum(), "of", n)
    edges = np.linspace(*RANGE, 7)
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (k >= lo) & (k <= hi)
        print(f"  {FEAT} [{lo:.2f},{hi:.2f}]: bad {np.mean(first_bad[sel] >= 0):.3f}  median first-bad step {np.median(first_bad[sel][first_bad[sel]>=0]) if (first_bad[sel]>=0).any() else -1}")
