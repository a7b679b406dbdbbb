# flake8: noqa: F401
from carl_amd.envs.gymnasium.classic_control.carl_acrobot import CARLAcrobot
from carl_amd.envs.gymnasium.classic_control.carl_cartpole import CARLCartPole
from carl_amd.envs.gymnasium.classic_control.carl_mountaincar import CARLMountainCar
from carl_amd.envs.gymnasium.classic_control.carl_mountaincarcontinuous import (
    CARLMountainCarContinuous,
)
from carl_amd.envs.gymnasium.classic_control.carl_pendulum import CARLPendulum

__all__ = [
    "CARLAcrobot",
    "CARLCartPole",
    "CARLMountainCar",
    "CARLMountainCarContinuous",
    "CARLPendulum",
]
