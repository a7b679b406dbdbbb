"""CARLCartPole: context-feature table of the reference (carl/envs/gymnasium/classic_control/carl_cartpole.py:11-66).

Only the feature table lives here.  The reset distribution the reference implements as a
Python ``reset()`` override -- state = U(initial_state_lower, initial_state_upper, size 4); obs = float32(state) (:44-66) --
and the step physics run in the HIP kernels of the ``CartPole-v1`` family
(carl_amd/csrc/classic_control.hip.h).
"""
from __future__ import annotations

import numpy as np

from carl_amd import spaces
from carl_amd.context.context_space import ContextFeature, UniformFloatContextFeature
from carl_amd.envs.gymnasium.carl_gymnasium_env import CARLGymnasiumEnv

# (name, lower, upper, default) in the reference's order = row order of the device table
_FEATURES = (
    ("gravity", 0.1, np.inf, 9.8),
    ("masscart", 0.1, 10, 1.0),
    ("masspole", 0.01, 1, 0.1),
    ("length", 0.05, 5, 0.5),
    ("force_mag", 1, 100, 10.0),
    ("tau", 0.002, 0.2, 0.02),
    ("initial_state_lower", -np.inf, np.inf, -0.1),
    ("initial_state_upper", -np.inf, np.inf, 0.1),
)


class CARLCartPole(CARLGymnasiumEnv):
    env_name: str = "CartPole-v1"
    metadata = {"render_modes": []}

    @staticmethod
    def get_context_features() -> dict[str, ContextFeature]:
        return {
            name: UniformFloatContextFeature(name, lower=lo, upper=hi, default_value=default)
            for name, lo, hi, default in _FEATURES
        }

    def _base_observation_space(self) -> spaces.Space:
        high = np.array([2.4 * 2, np.finfo(np.float32).max, 12 * 2 * np.pi / 360 * 2, np.finfo(np.float32).max],
                        dtype=np.float32)
        return spaces.Box(-high, high, dtype=np.float32)
