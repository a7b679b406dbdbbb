"""CARLBraxHopper: mirrors the reference's class (carl/envs/brax/carl_hopper.py:14-58).
Model: ``models.hopper_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxHopper(CARLBraxEnv):
    env_name = "hopper"
    asset_path = "envs/assets/hopper.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("hopper"))
