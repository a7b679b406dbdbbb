# flake8: noqa: F401
from carl_amd.envs.brax.carl_ant import CARLBraxAnt
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv

__all__ = ["CARLBraxEnv", "CARLBraxAnt"]
