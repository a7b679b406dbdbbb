"""CARLBraxAnt: mirrors the reference's class (carl/envs/brax/carl_ant.py:14-49).
The model itself (links, joints, colliders, motors) is ``models.ant_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxAnt(CARLBraxEnv):
    env_name = "ant"
    asset_path = "envs/assets/ant.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("ant"))
