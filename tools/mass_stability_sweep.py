"""Lowest stable mass ratio per Brax family and mass feature (explicit spring integration: k dt^2 / m_eff).
Sweeps every mass_<link> feature from 0.1 x to 1.0 x its default under a full-range random policy for 150 env
steps and reports the smallest ratio above which no env went non-finite / |obs| > 1e3.  Run on the GPU box."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from carl_amd.brax_engine import BraxVecEngine
from carl_amd.envs.brax.models import SYSTEMS
from oracle import oracle as O
from tools.brax_parity_percentiles import CLASSES
n = 2048
for fam in (sys.argv[1:] or list(CLASSES)):
    cls = CLASSES[fam]
    feats = cls.get_context_features(); names = list(feats)
    default = np.array([float(f.default_value) for f in feats.values()])
    s = SYSTEMS[cls.env_name](names)
    out = []
    for f in [nm for nm in names if nm.startswith("mass_")]:
        rows = np.tile(default, (n, 1))
        ratio = np.linspace(0.1, 1.0, n)
        rows[:, names.index(f)] = default[names.index(f)] * ratio
        eng = BraxVecEngine(s, len(names), rows, n, "cuda", max_episode_steps=10_000, auto_reset=False,
                            selector=O.SEL_STATIC, seed=5, ctx_idx0=np.arange(n))
        eng.reset()
        g = torch.Generator(device="cuda").manual_seed(0)
        lo, hi = float(min(s.act_lo[: s.n_act])), float(max(s.act_hi[: s.n_act]))
        bad = torch.zeros(n, dtype=torch.bool, device="cuda")
        for t in range(150):
            a = torch.rand((n, s.n_act), generator=g, device="cuda") * (hi - lo) + lo
            obs, rew, term, trunc = eng.step(a)
            bad |= ~torch.isfinite(obs).all(1) | (obs.abs().max(1).values > 1e3)
        b = bad.cpu().numpy()
        floor = float(ratio[b].max()) if b.any() else 0.0
        out.append((f, floor))
    print(f"{fam:26s}", "  ".join(f"{f}: {fl:.2f}" for f, fl in out), flush=True)
