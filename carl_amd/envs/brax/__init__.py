# flake8: noqa: F401
from carl_amd.envs.brax.carl_ant import CARLBraxAnt
from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv
from carl_amd.envs.brax.carl_halfcheetah import CARLBraxHalfcheetah
from carl_amd.envs.brax.carl_hopper import CARLBraxHopper
from carl_amd.envs.brax.carl_humanoid import CARLBraxHumanoid
from carl_amd.envs.brax.carl_humanoidstandup import CARLBraxHumanoidStandup
from carl_amd.envs.brax.carl_inverted_double_pendulum import CARLBraxInvertedDoublePendulum
from carl_amd.envs.brax.carl_pusher import CARLBraxPusher
from carl_amd.envs.brax.carl_reacher import CARLBraxReacher
from carl_amd.envs.brax.carl_inverted_pendulum import CARLBraxInvertedPendulum
from carl_amd.envs.brax.carl_walker2d import CARLBraxWalker2d

__all__ = ["CARLBraxEnv", "CARLBraxAnt", "CARLBraxHalfcheetah", "CARLBraxHumanoid", "CARLBraxHopper", "CARLBraxWalker2d",
           "CARLBraxInvertedPendulum", "CARLBraxHumanoidStandup", "CARLBraxInvertedDoublePendulum",
           "CARLBraxReacher", "CARLBraxPusher"]
