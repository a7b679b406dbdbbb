# flake8: noqa: F401
from carl_amd.envs.brax.carl_ant import CARLBraxAnt
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.carl_halfcheetah import CARLBraxHalfcheetah
from carl_amd.envs.brax.carl_humanoid import CARLBraxHumanoid

__all__ = ["CARLBraxEnv", "CARLBraxAnt", "CARLBraxHalfcheetah", "CARLBraxHumanoid"]
