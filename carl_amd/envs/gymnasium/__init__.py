# flake8: noqa: F401
from carl_amd.envs.gymnasium.carl_gymnasium_env import CARLGymnasiumEnv
from carl_amd.envs.gymnasium.classic_control import (
    CARLAcrobot,
    CARLCartPole,
    CARLMountainCar,
    CARLMountainCarContinuous,
    CARLPendulum,
)

__all__ = [
    "CARLGymnasiumEnv",
    "CARLAcrobot",
    "CARLCartPole",
    "CARLMountainCar",
    "CARLMountainCarContinuous",
    "CARLPendulum",
]
