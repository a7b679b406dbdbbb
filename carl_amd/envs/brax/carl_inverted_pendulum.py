"""CARLBraxInvertedPendulum: mirrors the reference's class (carl/envs/brax/carl_inverted_pendulum.py:9-38; no goal features).  Model:
``models.inverted_pendulum_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxInvertedPendulum(CARLBraxEnv):
    env_name = "inverted_pendulum"
    asset_path = "envs/assets/inverted_pendulum.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("inverted_pendulum"))
