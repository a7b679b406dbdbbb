"""Where does the explicit spring integration stay stable when `joint_stiffness` (this build's extension feature, BASELINE config 5)
scales the constraint stiffness?  Grid over (joint_stiffness scale, mass_torso ratio) for the two families that have the feature:
1 024 envs per cell, 300 env steps under uniform random actions from reset, auto-reset off; a cell counts the envs whose observation
left the finite / |x| < 1e4 range.  (GPU box; product engine only.)

    python tools/stiffness_stability_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402  (selector constants only)
from tests.test_gpu_brax import _cheetah, _humanoid  # noqa: E402


def main():
    from carl_amd.brax_engine import BraxVecEngine

    dev = torch.device("cuda", 0)
    n, T = 1024, 300
    S = [1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 6.0, 10.0]
    M = [0.8, 1.0, 1.5, 2.0, 3.0]
    for label, (s, names, default), amp in (("halfcheetah", _cheetah(), 1.0), ("humanoid", _humanoid(), 0.4)):
        names = list(names)
        print(f"{label}: envs of {n} that blew up within {T} steps; rows = joint_stiffness scale, columns = mass_torso ratio {M}")
        for sc in S:
            cells = []
            for mr in M:
                rows = np.tile(default, (n, 1))
                rows[:, names.index("joint_stiffness")] = sc
                rows[:, names.index("mass_torso")] = default[names.index("mass_torso")] * mr
                eng = BraxVecEngine(s, len(names), rows.astype(np.float32).astype(np.float64), n, dev, selector=O.SEL_STATIC, seed=5,
                                    ctx_idx0=np.arange(n), auto_reset=False, max_episode_steps=100_000)
                eng.reset()
                g = torch.Generator(device=dev).manual_seed(9)
                bad = torch.zeros(n, dtype=torch.bool, device=dev)
                for _ in range(T):
                    obs, *_ = eng.step((torch.rand((n, s.n_act), generator=g, device=dev) * 2 - 1) * amp)
                    bad |= ~torch.isfinite(obs).all(1) | (obs.abs().amax(1) > 1e4)
                cells.append(int(bad.sum()))
            print(f"    x {sc:5.1f}   " + "  ".join(f"{c:5d}" for c in cells))


if __name__ == "__main__":
    main()
