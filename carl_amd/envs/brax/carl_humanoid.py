"""CARLBraxHumanoid: mirrors the reference's class (carl/envs/brax/carl_humanoid.py:14-85; feature order preserved) + the ``joint_stiffness``
extension BASELINE config 5 asks for (SURVEY.md Quirk B4; appended, default 1).  Model:
``models.humanoid_sys`` (11 links, multi-dof waist / hip / shoulder joints, 244-dim obs)."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxHumanoid(CARLBraxEnv):
    env_name = "humanoid"
    asset_path = "envs/assets/humanoid.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("humanoid"))
