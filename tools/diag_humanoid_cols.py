import sys; sys.path.insert(0, ".")
import numpy as np, torch
from tools.brax_parity_percentiles import CLASSES, rel
from carl_amd.brax_engine import BraxVecEngine
from carl_amd.envs.brax.models import SYSTEMS
from oracle import brax as B, oracle as O
fam = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
cls = CLASSES[fam]; feats = cls.get_context_features(); names = list(feats)
default = np.array([float(f.default_value) for f in feats.values()])
s = SYSTEMS[cls.env_name](names); n = 2048; rng = np.random.default_rng(1)
rows = np.tile(default, (n, 1)).astype(np.float32).astype(np.float64)
kw = dict(selector=O.SEL_STATIC, seed=5, ctx_idx0=np.arange(n))
eng = BraxVecEngine(s, len(names), rows, n, "cuda", max_episode_steps=10_000, auto_reset=False, **kw)
ora = B.Engine(s, rows, n, max_steps=10_000, autoreset=False, **kw)
eng.reset(); ora.reset()
lo = np.array(s.act_lo[: s.n_act]); hi = np.array(s.act_hi[: s.n_act])
E = []
for t in range(30):
    ora.state[:] = eng.state_np()
    a = rng.uniform(lo, hi, (n, s.n_act)).astype(np.float32)
    obs, rew, term, trunc = eng.step(torch.as_tensor(a)); out = ora.step(a)
    E.append(rel(obs.cpu().numpy(), out.obs))
E = np.concatenate(E)  # [samples, D]
p99 = np.percentile(E, 99, axis=0)
nq = s.n_q - s.exclude_current_positions; nd = s.n_dof
blocks = {"q": (0, nq), "qd": (nq, nq + nd)}
if s.obs_extended:
    L = s.n_links; b0 = nq + nd
    blocks.update({"cinert": (b0, b0 + 10 * L), "cvel": (b0 + 10 * L, b0 + 16 * L), "qfrc": (b0 + 16 * L, b0 + 16 * L + nd)})
for k, (a0, a1) in blocks.items():
    print(f"{k:8s} cols [{a0},{a1}): p99 max over cols {p99[a0:a1].max():.2e}  median over cols {np.median(p99[a0:a1]):.2e}  worst col {a0 + int(p99[a0:a1].argmax())}")
