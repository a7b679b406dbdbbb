"""Family adapter for the classic-control envs (reference:
carl/envs/gymnasium/carl_gymnasium_env.py:19-77).

The reference builds ``gymnasium.make(env_name)`` and pushes contexts with
``setattr(env.unwrapped, k, v)``; here ``env_name`` selects a kernel family of the lane
engine and contexts live in a dense device table.  ``env=`` accepts a pre-built
``VecEngine`` -- the same injection hook the reference has (carl_gymnasium_env.py:25,63).
"""
from __future__ import annotations

import torch

from carl_amd import _lib, spaces
from carl_amd.context.selection import AbstractSelector
from carl_amd.engine import VecEngine
from carl_amd.envs.carl_env import CARLEnv
from carl_amd.utils.types import Contexts

# gymnasium registry id -> engine family
ENV_FAMILIES = {
    "CartPole-v1": _lib.CARTPOLE,
    "Pendulum-v1": _lib.PENDULUM,
    "Acrobot-v1": _lib.ACROBOT,
    "MountainCar-v0": _lib.MOUNTAINCAR,
    "MountainCarContinuous-v0": _lib.MOUNTAINCAR_CONT,
}


class CARLGymnasiumEnv(CARLEnv):
    env_name: str
    render_mode: str = "rgb_array"

    def __init__(
        self,
        env: VecEngine | None = None,
        contexts: Contexts | None = None,
        obs_context_features: list[str] | None = None,
        obs_context_as_dict: bool = True,
        context_selector: AbstractSelector | type[AbstractSelector] | None = None,
        context_selector_kwargs: dict = None,
        *,
        num_envs: int = 1,
        device: str | torch.device | None = None,
        auto_reset: bool | None = None,
        seed: int = 0,
        lane_offset: int = 0,
        context_offset: int | None = None,
        max_episode_steps: int | None = None,
        derived: str = "stale",
        fin_capacity: int = 0,
        **kwargs,
    ) -> None:
        """Reference parameters (carl_gymnasium_env.py:23-34) plus the batch ones:

        num_envs : lanes resident on ``device`` (1 = the reference's scalar API)
        auto_reset : reset done lanes inside ``step`` (default: True iff num_envs > 1)
        seed / lane_offset : Philox key and global id of lane 0 (multi-GPU sharding)
        context_offset : global id of row 0 of ``contexts`` when the table is a shard of a global context set
            (default: a table with one row per lane is this rank's slice of a lane <-> context identity,
            any other table is the whole set; VecEngine.default_ctx_idx)
        derived : CartPole only -- "stale" replicates the reference (Quirk C1:
            total_mass / polemass_length stay at gymnasium's init values), "recompute"
            derives them from the context.
        """
        if derived not in ("stale", "recompute"):
            raise ValueError("derived must be 'stale' or 'recompute'")
        if env is None:
            family = ENV_FAMILIES[self.env_name]
            F = len(self.get_context_features())
            env = VecEngine(
                family,
                [[float(cf.default_value) for cf in self.get_context_features().values()]],
                num_envs,
                device="cuda" if device is None else device,
                auto_reset=(num_envs > 1) if auto_reset is None else auto_reset,
                max_episode_steps=max_episode_steps,
                seed=seed,
                lane_offset=lane_offset,
                context_offset=context_offset,
                cartpole_recompute=(derived == "recompute"),
                fin_capacity=fin_capacity,
            )
            assert env.F == F, "context-feature table and kernel family disagree"
        super().__init__(
            env=env,
            contexts=contexts,
            obs_context_features=obs_context_features,
            obs_context_as_dict=obs_context_as_dict,
            context_selector=context_selector,
            context_selector_kwargs=context_selector_kwargs,
            **kwargs,
        )

    def _action_space(self) -> spaces.Space:
        info = self.env.info
        if info.action_is_discrete:
            return spaces.Discrete(info.n_actions)
        import numpy as np

        return spaces.Box(low=info.action_low, high=info.action_high, shape=(1,), dtype=np.float32)
