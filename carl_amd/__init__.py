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
    "CARLBraxAnt": "carl_amd.envs.brax",
    "CARLBraxHalfcheetah": "carl_amd.envs.brax",
    "CARLBraxHumanoid": "carl_amd.envs.brax",
    "CARLBraxHopper": "carl_amd.envs.brax",
    "CARLBraxWalker2d": "carl_amd.envs.brax",
    "CARLBraxInvertedPendulum": "carl_amd.envs.brax",
    "CARLBraxHumanoidStandup": "carl_amd.envs.brax",
    "CARLBraxInvertedDoublePendulum": "carl_amd.envs.brax",
    "CARLBraxReacher": "carl_amd.envs.brax",
    "CARLBraxPusher": "carl_amd.envs.brax",
    "CARLBraxHalfcheetahStiffness": "carl_amd.envs.brax",  # opt-in extension classes (joint_stiffness feature)
    "CARLBraxHumanoidStiffness": "carl_amd.envs.brax",
    "VecEngine": "carl_amd.engine",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        return getattr(importlib.import_module(_LAZY[name]), name)
    raise AttributeError(name)


# ---- registry: the ids the reference registers with gymnasium ("carl/<Cls>-v0",
# carl/__init__.py:31-35), resolvable without gymnasium -------------------------------
registry = {
    f"carl/{name}-v0": (module, name)
    for name, module in _LAZY.items()
    if name.startswith("CARL") and name != "CARLEnv" and not name.endswith("Stiffness")  # reference ids only
}


def make(env_id: str, **kwargs):
    """``gymnasium.make("carl/CARLCartPole-v0", contexts=...)`` equivalent."""
    if env_id not in registry:
        raise KeyError(f"unknown env id {env_id!r}; known: {sorted(registry)}")
    import importlib

    module, name = registry[env_id]
    return getattr(importlib.import_module(module), name)(**kwargs)
