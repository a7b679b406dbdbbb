"""CARLBraxHalfcheetah: context-feature table of the reference
(carl/envs/brax/carl_halfcheetah.py:14-67) + the ``joint_stiffness`` extension BASELINE config 5
asks for (SURVEY.md Quirk B4: the name exists only in the reference's legacy docs; here it
scales the spring backend's constraint stiffness, default 1).  Model: ``models.halfcheetah_sys``."""
from __future__ import annotations

import numpy as np

from carl_amd.context.context_space import (
    CategoricalContextFeature,
    ContextFeature,
    UniformFloatContextFeature,
)
from carl_amd.envs.brax.carl_ant import directions
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv

_MASSES = (("mass_torso", 10), ("mass_bthigh", 1.5435146), ("mass_bshin", 1.5874476), ("mass_bfoot", 1.0953975),
           ("mass_fthigh", 1.4380753), ("mass_fshin", 1.2008368), ("mass_ffoot", 0.8845188))


class CARLBraxHalfcheetah(CARLBraxEnv):
    env_name: str = "halfcheetah"
    asset_path: str = "envs/assets/half_cheetah.xml"
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
        for name, default in _MASSES:
            feats[name] = U(name, lower=1e-6, upper=np.inf, default_value=default)
        feats["target_distance"] = U("target_distance", lower=0, upper=np.inf, default_value=100)
        feats["target_direction"] = CategoricalContextFeature("target_direction", choices=directions, default_value=1)
        feats["target_radius"] = U("target_radius", lower=0.1, upper=np.inf, default_value=5)
        # extension, appended so that the reference's feature order is a prefix
        feats["joint_stiffness"] = U("joint_stiffness", lower=0.01, upper=100, default_value=1.0)
        return feats
