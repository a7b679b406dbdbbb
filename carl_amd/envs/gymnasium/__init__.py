"""gymnasium-shaped families: the base class and the classic-control envs."""
from carl_amd.envs.gymnasium import classic_control as _cc
from carl_amd.envs.gymnasium.carl_gymnasium_env import CARLGymnasiumEnv  # noqa: F401

__all__ = ["CARLGymnasiumEnv", *_cc.__all__]
globals().update({name: getattr(_cc, name) for name in _cc.__all__})
