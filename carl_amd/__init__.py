"""carl_amd -- MI355X-native batched step/reset engine behind CARL's env API.

``carl_amd.context`` (feature types, ContextSpace, ContextSampler, selectors) is pure
host code; ``carl_amd.envs`` / ``carl_amd.engine`` load the HIP library
(``carl_amd/lib/libcarl_amd.so``, built by ``python -m carl_amd.build``) and have no
CPU fallback.
"""
__version__ = "0.1.0"

_LAZY = {
    "CARLEnv": "carl_amd.envs.carl_env",
    "CARLCartPole": "carl_amd.envs.gymnasium.classic_control",
    "CARLPendulum": "carl_amd.envs.gymnasium.classic_control",
    "CARLAcrobot": "carl_amd.envs.gymnasium.classic_control",
    "CARLMountainCar": "carl_amd.envs.gymnasium.classic_control",
    "CARLMountainCarContinuous": "carl_amd.envs.gymnasium.classic_control",
    "VecEngine": "carl_amd.engine",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        return getattr(importlib.import_module(_LAZY[name]), name)
    raise AttributeError(name)
