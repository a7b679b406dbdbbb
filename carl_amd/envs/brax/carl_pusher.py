"""CARLBraxPusher: context-feature table of the reference (carl/envs/brax/carl_pusher.py:9-103).  The
three ``goal_position_*`` features are not physics: the reference hands them to the brax env as its
goal (``_update_context`` :91-103); here ``models._wire_context`` maps them to the goal rows the
kernel reads per env.  Model: ``models.pusher_sys``."""
from __future__ import annotations

import numpy as np

from carl_amd.context.context_space import ContextFeature, UniformFloatContextFeature
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.models import PUSHER_MASSES


class CARLBraxPusher(CARLBraxEnv):
    env_name: str = "pusher"
    asset_path: str = "envs/assets/pusher.xml"
    metadata = {"render_modes": []}
    task_context_features = ("goal_position_x", "goal_position_y", "goal_position_z")

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
        for name, value in PUSHER_MASSES.items():  # arm links, then the object to be pushed
            feats[name] = U(name, lower=1e-6, upper=np.inf, default_value=value)
        feats["goal_position_x"] = U("goal_position_x", lower=0, upper=np.inf, default_value=0.45)
        feats["goal_position_y"] = U("goal_position_y", lower=0, upper=np.inf, default_value=0.05)
        feats["goal_position_z"] = U("goal_position_z", lower=0, upper=np.inf, default_value=0.05)
        return feats
