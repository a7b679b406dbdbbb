"""Generate tests/golden/transitions_brax_<family>.npz from the float64 C restatement (oracle/brax_spring.c).

    python tests/golden/make_brax_transition_golden.py

48 rows per family (Ant, Halfcheetah, Humanoid -- BASELINE configs 4 and 5): (context row, state [L, 13], action) ->
(observation, reward, terminated, branch record) of ONE env step (= n_frames pipeline substeps), from states a
few random-policy steps away from reset, BASELINE-style context variation.  brax itself cannot produce these (not
importable here -- SURVEY.md 8c), so like tests/golden/transitions_<classic family>.npz the fixture is SELF-DERIVED: it
pins the oracle against drift (CPU test) and gives the HIP kernel a committed target (GPU test); it is not reference
output and says so wherever it is used.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carl_amd import envs as E  # noqa: E402
from carl_amd.envs.brax.models import SYSTEMS  # noqa: E402
from oracle import brax as B  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"ant": (E.CARLBraxAnt, {"gravity": (-15, -5), "friction": (0.3, 1.5), "mass_torso": (5, 15)}, 1.0),
         "halfcheetah": (E.CARLBraxHalfcheetahStiffness, {"joint_stiffness": (0.5, 2.0), "gravity": (-15, -5)}, 1.0),
         "humanoid": (E.CARLBraxHumanoidStiffness, {"joint_stiffness": (0.5, 2.0), "gravity": (-15, -5)}, 0.4)}
N = 48


def make(fam):
    cls, dists, amp = CASES[fam]
    feats = cls.get_context_features()
    names = list(feats)
    rng = np.random.default_rng(sum(map(ord, fam)))
    rows = np.tile([float(f.default_value) for f in feats.values()], (N, 1))
    for k, (lo, hi) in dists.items():
        rows[N // 3:, names.index(k)] = rng.uniform(lo, hi, N - N // 3)  # first third: the default context
    rows = rows.astype(np.float32).astype(np.float64)
    s = SYSTEMS[cls.env_name](names)
    eng = B.Engine(s, rows, N, selector=O.SEL_STATIC, ctx_idx0=np.arange(N), seed=7, autoreset=False, max_steps=1 << 30)
    eng.reset()
    for t in range(6):  # lanes k run k % 7 warm-up steps: reset poses, contacts, limits, motion
        a = (rng.uniform(-1, 1, (N, s.n_act)) * amp).astype(np.float32)
        keep = eng.state.copy()
        eng.step(a)
        still = (np.arange(N) % 7) <= t
        eng.state[still] = keep[still]
    # float32 head + tail representable states, as the engine holds them
    pose = eng.state.reshape(N, s.n_links, 13)
    head = pose[:, :, :7].astype(np.float32)
    tail = (pose[:, :, :7] - head).astype(np.float32)
    pose[:, :, :7] = head.astype(np.float64) + tail.astype(np.float64)
    pose[:, :, 7:] = pose[:, :, 7:].astype(np.float32)
    state = eng.state.copy()
    action = (rng.uniform(-1.2, 1.2, (N, s.n_act)) * amp).astype(np.float32)
    out = eng.step(action)
    np.savez_compressed(os.path.join(HERE, f"transitions_brax_{fam}.npz"), names=np.array(names), ctx=rows.astype(np.float32),
                        state=state, action=action, obs=out.obs, reward=out.reward,
                        terminated=out.terminated, branch_sig=eng.branch_sig)
    print(fam, "rows", N, "obs", out.obs.shape, "terminated", int(out.terminated.sum()), "with contacts",
          int((eng.branch_sig[:, 0] != 0).sum()))


if __name__ == "__main__":
    for fam in CASES:
        make(fam)
