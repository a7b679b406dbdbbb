"""CARLPendulum: context-feature table of the reference (carl/envs/gymnasium/classic_control/carl_pendulum.py:11-65).

Only the feature table lives here.  The reset distribution the reference implements as a
Python ``reset()`` override -- theta = U(0, initial_angle_max), thetadot = U(0, initial_velocity_max); obs = (cos, sin, thetadot) (:41-65).
    The feature named ``gravity`` is inert in the reference (Quirk P1); real gravity is ``g``. --
and the step physics run in the HIP kernels of the ``Pendulum-v1`` family
(carl_amd/csrc/classic_control.hip.h).
"""
from __future__ import annotations

import numpy as np

from carl_amd import spaces
from carl_amd.context.context_space import ContextFeature, UniformFloatContextFeature
from carl_amd.envs.gymnasium.carl_gymnasium_env import CARLGymnasiumEnv

# (name, lower, upper, default) in the reference's order = row order of the device table
_FEATURES = (
    ("gravity", -np.inf, np.inf, 8.0),
    ("dt", 0, np.inf, 0.05),
    ("g", 0, np.inf, 10),
    ("m", 1e-6, np.inf, 1),
    ("l", 1e-6, np.inf, 1),
    ("initial_angle_max", 0, np.inf, np.pi),
    ("initial_velocity_max", 0, np.inf, 1),
)


class CARLPendulum(CARLGymnasiumEnv):
    env_name: str = "Pendulum-v1"
    metadata = {"render_modes": []}

    @staticmethod
    def get_context_features() -> dict[str, ContextFeature]:
        return {
            name: UniformFloatContextFeature(name, lower=lo, upper=hi, default_value=default)
            for name, lo, hi, default in _FEATURES
        }

    def _base_observation_space(self) -> spaces.Space:
        high = np.array([1.0, 1.0, 8.0], dtype=np.float32)
        return spaces.Box(-high, high, dtype=np.float32)
