"""CARLBraxHumanoidStandup: context-feature table of the reference
(carl/envs/brax/carl_humanoidstandup.py:9-71: the Humanoid features without goal features, and
``ang_damping`` default -0.05).  Model: ``models.humanoidstandup_sys``."""
from __future__ import annotations

import numpy as np

from carl_amd.context.context_space import ContextFeature, UniformFloatContextFeature
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.models import HUMANOID_MASSES


class CARLBraxHumanoidStandup(CARLBraxEnv):
    env_name: str = "humanoidstandup"
    asset_path: str = "envs/assets/humanoidstandup.xml"
    metadata = {"render_modes": []}

    @staticmethod
    def get_context_features() -> dict[str, ContextFeature]:
        U = UniformFloatContextFeature
        feats = {
            "gravity": U("gravity", lower=-1000, upper=-1e-6, default_value=-9.8),
            "friction": U("friction", lower=0, upper=100, default_value=1),
            "elasticity": U("elasticity", lower=0, upper=100, default_value=0),
            "ang_damping": U("ang_damping", lower=-np.inf, upper=np.inf, default_value=-0.05),
            "viscosity": U("viscosity", lower=0, upper=np.inf, default_value=0),
        }
        for name, default in HUMANOID_MASSES.items():
            feats[name] = U(name, lower=1e-6, upper=np.inf, default_value=default)
        return feats
