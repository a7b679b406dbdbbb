"""CARLBraxReacher: mirrors the reference's class (carl/envs/brax/carl_reacher.py:9-39).
Model: ``models.reacher_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxReacher(CARLBraxEnv):
    env_name = "reacher"
    asset_path = "envs/assets/reacher.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("reacher"))
