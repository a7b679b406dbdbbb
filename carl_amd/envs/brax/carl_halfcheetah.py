"""CARLBraxHalfcheetah: mirrors the reference's class (carl/envs/brax/carl_halfcheetah.py:14-67; feature order preserved) -- same context features, same
default context and observation space.  ``CARLBraxHalfcheetahStiffness`` is this build's opt-in variant with one extra, LAST
context feature, ``joint_stiffness`` (scale of the spring backend's constraint stiffness, default 1): BASELINE config 5
asks for "joint_stiffness variation", a name that exists only in the reference's legacy docs (SURVEY.md Quirk B4).
Model: ``models.halfcheetah_sys``."""
from __future__ import annotations

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.feature_tables import feature_table


class CARLBraxHalfcheetah(CARLBraxEnv):
    env_name = "halfcheetah"
    asset_path = "envs/assets/half_cheetah.xml"
    metadata = {"render_modes": []}
    get_context_features = staticmethod(lambda: feature_table("halfcheetah"))


class CARLBraxHalfcheetahStiffness(CARLBraxHalfcheetah):
    get_context_features = staticmethod(lambda: feature_table("halfcheetah", extensions=True))

