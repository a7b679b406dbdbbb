"""CARLBraxAnt: context-feature table of the reference (carl/envs/brax/carl_ant.py:14-49).
The model itself (links, joints, colliders, motors) is ``models.ant_sys``."""
from __future__ import annotations

import numpy as np

from carl_amd.context.context_space import (
    CategoricalContextFeature,
    ContextFeature,
    UniformFloatContextFeature,
)
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv

# carl/envs/brax/brax_walker_goal_wrapper.py:33-50
directions = [1, 3, 2, 4, 12, 32, 14, 34, 112, 332, 114, 334, 212, 232, 414, 434]


class CARLBraxAnt(CARLBraxEnv):
    env_name: str = "ant"
    asset_path: str = "envs/assets/ant.xml"
    metadata = {"render_modes": []}

    @staticmethod
    def get_context_features() -> dict[str, ContextFeature]:
        U = UniformFloatContextFeature
        return {
            "gravity": U("gravity", lower=-1000, upper=-1e-6, default_value=-9.8),
            "friction": U("friction", lower=0, upper=100, default_value=1),
            "elasticity": U("elasticity", lower=0, upper=100, default_value=0),
            "ang_damping": U("ang_damping", lower=-np.inf, upper=np.inf, default_value=-0.05),
            "mass_torso": U("mass_torso", lower=1e-6, upper=np.inf, default_value=10),
            "viscosity": U("viscosity", lower=0, upper=np.inf, default_value=0),
            "target_distance": U("target_distance", lower=0, upper=np.inf, default_value=100),
            "target_direction": CategoricalContextFeature("target_direction", choices=directions, default_value=1),
            "target_radius": U("target_radius", lower=0.1, upper=np.inf, default_value=5),
        }
