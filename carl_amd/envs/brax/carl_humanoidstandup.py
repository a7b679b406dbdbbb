"""CARLBraxHumanoidStandup: mirrors the reference's class (carl/envs/brax/carl_humanoidstandup.py:9-71: the Humanoid features without goal features, and
``ang_damping`` default -0.05).  Model: ``models.humanoidstandup_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxHumanoidStandup(CARLBraxEnv):
    env_name = "humanoidstandup"
    asset_path = "envs/assets/humanoidstandup.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("humanoidstandup"))
