"""CARLBraxInvertedDoublePendulum: context-feature table of the reference
(carl/envs/brax/carl_inverted_double_pendulum.py:9-41).  The reference registers the second pole's
mass under the KEY ``mass_pole2`` but with the feature NAME ``mass_pole`` (:32-34); the key is what the
context dict carries, so the feature is called ``mass_pole2`` here.  Model:
``models.inverted_double_pendulum_sys``."""
from __future__ import annotations

import numpy as np

from carl_amd.context.context_space import ContextFeature, UniformFloatContextFeature
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv


class CARLBraxInvertedDoublePendulum(CARLBraxEnv):
    env_name: str = "inverted_double_pendulum"
    asset_path: str = "envs/assets/inverted_double_pendulum.xml"
    metadata = {"render_modes": []}

    @staticmethod
    def get_context_features() -> dict[str, ContextFeature]:
        U = UniformFloatContextFeature
        return {
            "gravity": U("gravity", lower=-1000, upper=-1e-6, default_value=-9.8),
            "friction": U("friction", lower=0, upper=100, default_value=1),
            "elasticity": U("elasticity", lower=0, upper=100, default_value=0),
            "mass_cart": U("mass_cart", lower=1e-6, upper=np.inf, default_value=1),
            "mass_pole": U("mass_pole", lower=1e-6, upper=np.inf, default_value=1),
            "mass_pole2": U("mass_pole2", lower=1e-6, upper=np.inf, default_value=1),
            "ang_damping": U("ang_damping", lower=-np.inf, upper=np.inf, default_value=-0.05),
            "viscosity": U("viscosity", lower=0, upper=np.inf, default_value=0),
        }
