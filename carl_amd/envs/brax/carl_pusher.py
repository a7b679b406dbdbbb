"""CARLBraxPusher: mirrors the reference's class (carl/envs/brax/carl_pusher.py:9-103).  The
three ``goal_position_*`` features are not physics: the reference hands them to the brax env as its
goal (``_update_context`` :91-103); here ``models._wire_context`` maps them to the goal rows the
kernel reads per env.  Model: ``models.pusher_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxPusher(CARLBraxEnv):
    env_name = "pusher"
    asset_path = "envs/assets/pusher.xml"
    metadata = {"render_modes": []}
    task_context_features = ("goal_position_x", "goal_position_y", "goal_position_z")
    get_context_features = staticmethod(lambda: feature_table("pusher"))
