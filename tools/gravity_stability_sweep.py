"""Which single physics feature, pushed to the reference's declared bound with everything else at its default, makes a Brax class leave the
finite range?  python tools/gravity_stability_sweep.py [gravity | friction | elasticity]   (512 envs x 300 steps per cell, random actions; GPU box)"""
import os, sys, inspect
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import oracle as O
import carl_amd.envs.brax as be
from carl_amd.brax_engine import BraxVecEngine
from carl_amd.envs.brax.models import SYSTEMS
dev = torch.device("cuda", 0); n = 512; T = 300
FEAT = sys.argv[1] if len(sys.argv) > 1 else "gravity"
G = {"gravity": [-50, -100, -200, -500, -1000], "friction": [3, 10, 30, 100], "elasticity": [0.9, 1.0, 2.0, 10.0, 100.0]}[FEAT]
for cname, cls in inspect.getmembers(be):
    if not (inspect.isclass(cls) and cname.startswith("CARLBrax") and cname != "CARLBraxEnv") or cname.endswith("Stiffness"): continue
    feats = cls.get_context_features(); names = list(feats); default = np.array([float(f.default_value) for f in feats.values()])
    s = SYSTEMS[cls.env_name](names); amp = 0.4 if "humanoid" in cls.env_name else 1.0
    cells = []
    for g0 in G:
        rows = np.tile(default, (n, 1)); rows[:, names.index(FEAT)] = g0
        eng = BraxVecEngine(s, len(names), rows, n, dev, selector=O.SEL_STATIC, seed=5, ctx_idx0=np.arange(n), auto_reset=False, max_episode_steps=100000)
        eng.reset(); g = torch.Generator(device=dev).manual_seed(9); bad = torch.zeros(n, dtype=torch.bool, device=dev)
        for _ in range(T):
            obs, *_ = eng.step((torch.rand((n, s.n_act), generator=g, device=dev) * 2 - 1) * amp)
            bad |= ~torch.isfinite(obs).all(1) | (obs.abs().amax(1) > 1e4)
        cells.append(int(bad.sum()))
    print(f"{cname:32s} blown up of {n} at {FEAT} {G}: {cells}")
