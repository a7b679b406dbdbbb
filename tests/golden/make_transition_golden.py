"""Generate tests/golden/transitions_<family>.npz from the float64 CPU oracle.

    python tests/golden/make_transition_golden.py

1024 rows per family: (ctx row, state, action) -> (state', obs, reward, terminated),
random contexts inside the reference's feature bounds, random states over the reachable
range, plus rows parked next to every termination threshold.  The reference itself
cannot produce these (gymnasium is not importable here -- SURVEY.md section 8c), so the
fixture pins the ORACLE against drift and gives the HIP path a committed target; it is
"self-derived", not reference output, and is labelled so in tests and DESIGN.md.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
N = 1024


def f32(x):
    """inputs are float32-representable so fp32 and fp64 paths start from equal values"""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def contexts(fam, rng, n):
    c = np.tile(O.default_row(fam), (n, 1))
    half = n // 2  # first half default context, second half varied
    v = c[half:]
    m = v.shape[0]
    U = rng.uniform
    if fam == O.CARTPOLE:
        v[:, 0] = U(1, 20, m); v[:, 1] = U(0.5, 3, m); v[:, 2] = U(0.05, 0.5, m)
        v[:, 3] = U(0.2, 2, m); v[:, 4] = U(2, 30, m); v[:, 5] = U(0.005, 0.05, m)
    elif fam == O.PENDULUM:
        v[:, 0] = U(1, 20, m); v[:, 1] = U(0.01, 0.1, m); v[:, 2] = U(1, 20, m)
        v[:, 3] = U(0.3, 3, m); v[:, 4] = U(0.5, 2, m)
    elif fam == O.ACROBOT:
        v[:, 0] = U(0.5, 2, m); v[:, 2] = U(0.5, 2, m); v[:, 3] = U(0.5, 2, m)
        v[:, 4] = U(0.3, 0.7, m); v[:, 5] = U(0.3, 0.7, m); v[:, 6] = U(0.5, 2, m)
        v[:, 7] = U(2 * np.pi, 6 * np.pi, m); v[:, 8] = U(4 * np.pi, 12 * np.pi, m)
    elif fam == O.MOUNTAINCAR:
        v[:, 2] = U(0.04, 0.1, m); v[:, 3] = U(0.3, 0.55, m); v[:, 5] = U(5e-4, 2e-3, m)
        v[:, 6] = U(1.5e-3, 3.5e-3, m)
    else:
        v[:, 2] = U(0.04, 0.1, m); v[:, 3] = U(0.3, 0.55, m); v[:, 5] = U(5e-4, 3e-3, m)
    return f32(c)


def states_actions(fam, rng, n):
    U = rng.uniform
    if fam == O.CARTPOLE:
        s = np.stack([U(-2.6, 2.6, n), U(-3, 3, n), U(-0.25, 0.25, n), U(-3, 3, n)], 1)
        k = n // 8  # rows hugging the x and theta thresholds
        s[:k, 0] = np.sign(U(-1, 1, k)) * (2.4 + U(-2e-3, 2e-3, k))
        s[k:2 * k, 2] = np.sign(U(-1, 1, k)) * (12 * 2 * np.pi / 360 + U(-2e-4, 2e-4, k))
        a = rng.integers(0, 2, n)
    elif fam == O.PENDULUM:
        s = np.stack([U(-12, 12, n), U(-8, 8, n)], 1)
        a = U(-3, 3, n)
    elif fam == O.ACROBOT:
        s = np.stack([U(-np.pi, np.pi, n), U(-np.pi, np.pi, n), U(-4 * np.pi, 4 * np.pi, n),
                      U(-9 * np.pi, 9 * np.pi, n)], 1)
        a = rng.integers(0, 3, n)
    else:
        s = np.stack([U(-1.2, 0.6, n), U(-0.07, 0.07, n)], 1)
        k = n // 8
        s[:k, 0] = -1.2 + U(0, 0.02, k)      # left wall
        s[k:2 * k, 0] = 0.45 + U(-0.03, 0.06, k)  # goal region
        a = rng.integers(0, 3, n) if fam == O.MOUNTAINCAR else U(-1.5, 1.5, n)
    return f32(s), a


def main():
    for fam in range(5):
        rng = np.random.default_rng(1000 + fam)
        ctx = contexts(fam, rng, N)
        s, a = states_actions(fam, rng, N)
        if fam in O.CONTINUOUS:
            a = np.asarray(a, dtype=np.float32)
        else:
            a = np.asarray(a, dtype=np.int32)
        s2, obs, rew, term = O.transitions(fam, ctx, s, a, precision="f64")
        path = os.path.join(HERE, f"transitions_{O.FAMILY_NAMES[fam]}.npz")
        np.savez_compressed(path, ctx=ctx.astype(np.float32), state=s.astype(np.float32), action=a,
                            next_state=s2, obs=obs, reward=rew, terminated=term)
        print(path, os.path.getsize(path), "bytes; terminated rows:", int(term.sum()))


if __name__ == "__main__":
    main()
