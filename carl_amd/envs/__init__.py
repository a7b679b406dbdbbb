# flake8: noqa: F401
from carl_amd.envs.brax import *  # noqa: F403
from carl_amd.envs.brax import __all__ as _brax_all
from carl_amd.envs.carl_env import CARLEnv
from carl_amd.envs.gymnasium import *  # noqa: F403
from carl_amd.envs.gymnasium import __all__ as _gym_all

__all__ = ["CARLEnv", *_gym_all, *_brax_all]
