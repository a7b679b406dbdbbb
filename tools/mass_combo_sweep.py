"""Do the single-feature mass floors (feature_tables.MASS_RATIO_FLOOR) hold when SEVERAL links are light at once?
Every mass feature of a family is put at alpha x its floor simultaneously (no clamp: the sys table is built
directly), alpha swept over the lanes; reports the smallest alpha above which no env went non-finite under a
full-range random policy for 150 env steps.  Run on the GPU box:  python tools/mass_combo_sweep.py [family ...]"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from carl_amd.brax_engine import BraxVecEngine
from carl_amd.envs.brax.feature_tables import DEFAULT_MASS_RATIO_FLOOR, MASS_RATIO_FLOOR
from carl_amd.envs.brax.models import SYSTEMS
from oracle import oracle as O
from tools.brax_parity_percentiles import CLASSES
n = 2048
for fam in (sys.argv[1:] or list(CLASSES)):
    cls = CLASSES[fam]
    feats = cls.get_context_features(); names = list(feats)
    default = np.array([float(f.default_value) for f in feats.values()])
    s = SYSTEMS[cls.env_name](names)
    floors = MASS_RATIO_FLOOR.get(cls.env_name, {})
    alpha = np.linspace(1.0, 3.0, n)
    rows = np.tile(default, (n, 1))
    for f in [nm for nm in names if nm.startswith("mass_")]:
        j = names.index(f)
        rows[:, j] = default[j] * np.minimum(1.0, alpha * floors.get(f, DEFAULT_MASS_RATIO_FLOOR))
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
    print(f"{fam:26s} all mass features at alpha x floor: unstable up to alpha = {float(alpha[b].max()) if b.any() else 0.0:.2f} "
          f"({int(b.sum())} of {n} lanes)", flush=True)
