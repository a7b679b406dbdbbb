"""CARLBraxHumanoid: mirrors the reference's class (carl/envs/brax/carl_humanoid.py:14-85; feature order preserved) -- same context features, same
default context and observation space.  ``CARLBraxHumanoidStiffness`` is this build's opt-in variant with one extra, LAST
context feature, ``joint_stiffness`` (scale of the spring backend's constraint stiffness, default 1): BASELINE config 5
asks for "joint_stiffness variation", a name that exists only in the reference's legacy docs (SURVEY.md Quirk B4).
Model: ``models.humanoid_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxHumanoid(CARLBraxEnv):
    env_name = "humanoid"
    asset_path = "envs/assets/humanoid.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("humanoid"))


class CARLBraxHumanoidStiffness(CARLBraxHumanoid):
    get_context_features = staticmethod(lambda: feature_table("humanoid", extensions=True))

