"""CARLBraxInvertedDoublePendulum: mirrors the reference's class (carl/envs/brax/carl_inverted_double_pendulum.py:9-41).  The reference registers the second pole's
mass under the KEY ``mass_pole2`` but with the feature NAME ``mass_pole`` (:32-34); the key is what the
context dict carries, so the feature is called ``mass_pole2`` here.  Model:
``models.inverted_double_pendulum_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxInvertedDoublePendulum(CARLBraxEnv):
    env_name = "inverted_double_pendulum"
    asset_path = "envs/assets/inverted_double_pendulum.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("inverted_double_pendulum"))
