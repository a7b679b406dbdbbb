"""CARLBraxWalker2d: context-feature table of the reference (carl/envs/brax/carl_walker2d.py:14-67).
Model: ``models.walker2d_sys``."""
from __future__ import annotations

from carl_amd.context.context_space import ContextFeature
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.carl_hopper import _walker_features


class CARLBraxWalker2d(CARLBraxEnv):
    env_name: str = "walker2d"
    asset_path: str = "envs/assets/walker2d.xml"
    metadata = {"render_modes": []}

    @staticmethod
    def get_context_features() -> dict[str, ContextFeature]:
        return _walker_features((("mass_torso", 10), ("mass_thigh", 4.0578904), ("mass_leg", 2.7813568),
                                 ("mass_foot", 3.1667254), ("mass_thigh_left", 4.0578904),
                                 ("mass_leg_left", 2.7813568), ("mass_foot_left", 3.1667254)))
