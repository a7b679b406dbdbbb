"""The ten Brax families (module and class names as in the reference package) and their base class."""
import importlib

from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv  # noqa: F401

_FAMILIES = {
    "ant": "Ant", "halfcheetah": "Halfcheetah", "humanoid": "Humanoid", "hopper": "Hopper", "walker2d": "Walker2d",
    "inverted_pendulum": "InvertedPendulum", "humanoidstandup": "HumanoidStandup",
    "inverted_double_pendulum": "InvertedDoublePendulum", "reacher": "Reacher", "pusher": "Pusher",
}
__all__ = ["CARLBraxEnv"]
for _mod, _suffix in _FAMILIES.items():
    _cls = "CARLBrax" + _suffix
    globals()[_cls] = getattr(importlib.import_module(f"{__name__}.carl_{_mod}"), _cls)
    __all__.append(_cls)
# this build's opt-in variants with the joint_stiffness context feature (BASELINE config 5; not reference classes)
from carl_amd.envs.brax.carl_halfcheetah import CARLBraxHalfcheetahStiffness  # noqa: E402
from carl_amd.envs.brax.carl_humanoid import CARLBraxHumanoidStiffness  # noqa: E402

__all__ += ["CARLBraxHalfcheetahStiffness", "CARLBraxHumanoidStiffness"]
del _mod, _suffix, _cls
