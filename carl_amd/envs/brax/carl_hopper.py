"""CARLBraxHopper: context-feature table of the reference (carl/envs/brax/carl_hopper.py:14-58).
Model: ``models.hopper_sys``."""
from __future__ import annotations

import numpy as np

from carl_amd.context.context_space import CategoricalContextFeature, ContextFeature, UniformFloatContextFeature
from carl_amd.envs.brax.carl_ant import directions
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv


def _walker_features(masses) -> dict[str, ContextFeature]:
    U = UniformFloatContextFeature
    feats = {
        "gravity": U("gravity", lower=-1000, upper=-1e-6, default_value=-9.8),
        "friction": U("friction", lower=0, upper=100, default_value=1),
        "elasticity": U("elasticity", lower=0, upper=100, default_value=0),
        "ang_damping": U("ang_damping", lower=-np.inf, upper=np.inf, default_value=-0.05),
        "viscosity": U("viscosity", lower=0, upper=np.inf, default_value=0),
    }
    for name, default in masses:
        feats[name] = U(name, lower=1e-6, upper=np.inf, default_value=default)
    feats["target_distance"] = U("target_distance", lower=0, upper=np.inf, default_value=100)
    feats["target_direction"] = CategoricalContextFeature("target_direction", choices=directions, default_value=1)
    feats["target_radius"] = U("target_radius", lower=0.1, upper=np.inf, default_value=5)
    return feats


class CARLBraxHopper(CARLBraxEnv):
    env_name: str = "hopper"
    asset_path: str = "envs/assets/hopper.xml"
    metadata = {"render_modes": []}

    @staticmethod
    def get_context_features() -> dict[str, ContextFeature]:
        return _walker_features((("mass_torso", 10), ("mass_thigh", 4.0578904), ("mass_leg", 2.7813568),
                                 ("mass_foot", 5.3155746)))
