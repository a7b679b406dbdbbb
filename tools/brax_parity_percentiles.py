"""Per-env-step parity of the HIP Brax kernel against the fp64 restatement (oracle/brax_spring.c) for every
Brax family: the oracle restarts every env step from the engine's float32 state (so only the arithmetic of
ONE env step = n_frames substeps is compared), random actions, BASELINE-style context variation.
Lanes are classified by their DISCRETE decisions (tests/brax_parity_util.py: contact set per substep, hashed on
both sides; `terminated`): prints percentiles and the MAXIMUM of |d| / (1 + |x|) over observation entries and
reward on the agreeing lanes, the excluded shares, and the all-lanes figures round 2 reported.  Run on the GPU box:
    python tools/brax_parity_percentiles.py [family ...]"""
import os
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from carl_amd import envs as E  # noqa: E402
from carl_amd.brax_engine import BraxVecEngine  # noqa: E402
from carl_amd.envs.brax.models import SYSTEMS  # noqa: E402
from oracle import brax as B  # noqa: E402
from oracle import oracle as O  # noqa: E402

CLASSES = {"ant": E.CARLBraxAnt, "halfcheetah": E.CARLBraxHalfcheetahStiffness, "humanoid": E.CARLBraxHumanoid,
           "hopper": E.CARLBraxHopper, "walker2d": E.CARLBraxWalker2d,
           "inverted_pendulum": E.CARLBraxInvertedPendulum,
           "inverted_double_pendulum": E.CARLBraxInvertedDoublePendulum,
           "humanoidstandup": E.CARLBraxHumanoidStandup, "reacher": E.CARLBraxReacher, "pusher": E.CARLBraxPusher}


sys.path.insert(0, "tests")
from brax_parity_util import Parity, step_both  # noqa: E402


def measure(fam, n=2048, steps=40, seed=1):
    cls = CLASSES[fam]
    feats = cls.get_context_features()
    names = list(feats)
    default = np.array([float(f.default_value) for f in feats.values()])
    s = SYSTEMS[cls.env_name](names)
    rng = np.random.default_rng(seed)
    rows = np.tile(default, (n, 1))
    for name, (lo, hi) in {"gravity": (-15, -5), "friction": (0.3, 1.5), "mass_torso": (5, 15),
                           "joint_stiffness": (0.5, 2.0)}.items():
        if name in names and fam not in ("pusher", "reacher") and not (fam == "halfcheetah" and name == "mass_torso"):
            rows[:, names.index(name)] = rng.uniform(lo, hi, n)
    rows = rows.astype(np.float32).astype(np.float64)
    kw = dict(selector=O.SEL_STATIC, seed=5, ctx_idx0=np.arange(n))
    fp32 = os.environ.get("CARL_BRAX_FP32") == "1" and not (s.target_link > 0 or s.push_link > 0)  # opt-in float32 substeps
    eng = BraxVecEngine(s, len(names), rows, n, "cuda", max_episode_steps=10_000, auto_reset=False, branch_record=True,
                        pose_float32=fp32, **kw)
    if fp32:
        fam = fam + " [float32 substeps]"
    ora = B.Engine(s, rows, n, max_steps=10_000, autoreset=False, **kw)
    eng.reset()
    ora.reset()
    lo = np.array(s.act_lo[: s.n_act]) * 1.2
    hi = np.array(s.act_hi[: s.n_act]) * 1.2
    par = Parity()
    for t in range(steps):
        a = rng.uniform(lo, hi, (n, s.n_act)).astype(np.float32)
        step_both(eng, ora, a, par, t)
    print(par.summary(f"{fam} ({s.n_frames} substeps)"), "worst agreeing (err, step, lane, col):", par.worst, flush=True)


if __name__ == "__main__":
    for fam in (sys.argv[1:] or list(CLASSES)):
        measure(fam)
