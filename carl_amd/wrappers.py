"""Adapters that let the documented user flows of the reference run on the lane engine
(reference: examples/carl_with_sb3.py:22-36 wraps a CARL env in
``gymnasium.wrappers.FlattenObservation`` before handing it to SB3).

``FlattenObservation`` here does the same flattening -- ``{"obs": o, "context": c}`` ->
``concat(o, c)`` -- on device tensors for batched envs and on NumPy for the scalar API, in
gymnasium's order for Dict spaces (keys sorted alphabetically: "context" before "obs")."""
from __future__ import annotations

import numpy as np
import torch

from carl_amd import spaces


class FlattenObservation:
    def __init__(self, env):
        self.env = env
        self.num_envs = getattr(env, "num_envs", 1)
        self.action_space = env.action_space
        self._ctx_names = list(env.obs_context_features)
        single = env.single_observation_space
        n_ctx = len(self._ctx_names)
        obs_box = single["obs"]
        # context bounds from the env's context space (gymnasium flattens the Dict's "context" Box, which carries
        # the feature bounds: carl/context/context_space.py:166-188), in the flattened (sorted-key) order
        feats = env.get_context_features()
        order = sorted(self._ctx_names) if env.obs_context_as_dict else self._ctx_names

        def bound(name, which, default):
            v = getattr(feats.get(name), which, None)
            return default if v is None else float(v)

        c_lo = np.array([bound(k, "lower", -np.inf) for k in order], np.float32)
        c_hi = np.array([bound(k, "upper", np.inf) for k in order], np.float32)
        lo = np.concatenate([c_lo, np.asarray(obs_box.low, np.float32).reshape(-1)])
        hi = np.concatenate([c_hi, np.asarray(obs_box.high, np.float32).reshape(-1)])
        # lanes move to other contexts on reset unless the selector is static: the TERMINAL observation of a
        # finished lane belongs to the context of the episode that ended, i.e. the one before the step
        from carl_amd import _lib

        self._ctx_moves = (not env._scalar_api) and int(env.env.b.selector) not in (_lib.SEL_STATIC, _lib.SEL_HOST)
        self.single_observation_space = spaces.Box(lo, hi, dtype=np.float32)
        self.observation_space = (self.single_observation_space if self.num_envs == 1 or env._scalar_api
                                  else spaces.batch_space(self.single_observation_space, self.num_envs))

    def _flat(self, obs):
        ctx, o = obs["context"], obs["obs"]
        if torch.is_tensor(o):
            c = torch.stack([ctx[k] for k in sorted(ctx)], dim=1) if isinstance(ctx, dict) else ctx
            return torch.cat([c, o], dim=1)
        c = [ctx[k] for k in sorted(ctx)] if isinstance(ctx, dict) else list(ctx)
        return np.concatenate([np.asarray(c, dtype=np.float32), np.asarray(o, dtype=np.float32).reshape(-1)])

    def reset(self, **kw):
        obs, info = self.env.reset(**kw)
        return self._flat(obs), info

    def _ctx_matrix(self, ctx):
        return torch.stack([ctx[k] for k in sorted(ctx)], dim=1) if isinstance(ctx, dict) else ctx

    def step(self, action):
        # (ctx_obs aliases a live engine buffer that the step rewrites for lanes that reset onto another context)
        before = self._ctx_matrix(self.env._batched_obs()["context"]).clone() if self._ctx_moves else None
        obs, reward, terminated, truncated, info = self.env.step(action)
        if "final_observation" in info and torch.is_tensor(info["final_observation"]):
            info = dict(info)  # terminal observation of done lanes, flattened the same way
            c = before if before is not None else self._ctx_matrix(obs["context"])
            info["final_observation"] = torch.cat([c, info["final_observation"]], dim=1)
        return self._flat(obs), reward, terminated, truncated, info

    def __getattr__(self, name):
        return getattr(self.env, name)


class SB3VecEnv:
    """stable-baselines3 ``VecEnv``-shaped facade over a batched CARL env (SURVEY.md section 8f rank 4;
    reference flow: examples/carl_with_sb3.py:22-36 = ``FlattenObservation`` -> ``DummyVecEnv`` ->
    ``PPO``).  SB3's protocol is NumPy on the host: ``reset() -> obs``, ``step_async(actions)`` /
    ``step_wait() -> (obs, rewards, dones, infos)`` with ``infos[i]["terminal_observation"]`` and
    ``infos[i]["TimeLimit.truncated"]`` for envs that finished, the returned obs being the first
    observation of the next episode.  That is the engine's auto-reset contract, so the facade only
    moves tensors to the host (one device->host copy per output per step) and fills info dicts for
    the finished envs (ballot-compacted ``carl_done_compact`` list, not a Python scan of N flags).
    stable-baselines3 is not installed in this image: the class is duck-typed (no base class), which
    SB3's algorithms accept through ``VecEnvWrapper``-style attribute access."""

    _EMPTY: dict = {}

    def __init__(self, env, flatten: bool = True):
        if getattr(env, "_scalar_api", False) or env.num_envs < 2:
            raise ValueError("SB3VecEnv wraps a batched env (num_envs > 1)")
        self.env = FlattenObservation(env) if flatten else env
        self.carl_env = env
        self.num_envs = env.num_envs
        self.observation_space = self.env.single_observation_space
        self.action_space = env.single_action_space
        self._actions = None
        self.reset_infos = [dict() for _ in range(self.num_envs)]

    # ---- VecEnv protocol -------------------------------------------------------------
    def reset(self):
        obs, _ = self.env.reset()
        return self._host(obs)

    def seed(self, seed=None):
        if seed is not None:
            self.carl_env.env.seed(int(seed))
        return [seed] * self.num_envs

    def step_async(self, actions) -> None:
        self._actions = actions

    def step_wait(self):
        eng = self.carl_env.env
        a = torch.as_tensor(np.asarray(self._actions), device=eng.device)
        obs, reward, term, trunc, info = self.env.step(a)
        done = term | trunc
        infos = [self._EMPTY] * self.num_envs
        idx, count = eng.done_compact()  # ascending ids of finished envs, built on the device
        idx = idx[: int(count.item())]
        if idx.numel():
            ids = idx.cpu().numpy()
            final = info["final_observation"][idx.long()].cpu().numpy()
            tr = trunc[idx.long()].cpu().numpy()
            te = term[idx.long()].cpu().numpy()
            for k, i in enumerate(ids):
                infos[i] = {"terminal_observation": final[k], "TimeLimit.truncated": bool(tr[k] and not te[k])}
        return self._host(obs), reward.cpu().numpy(), done.cpu().numpy(), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        self.carl_env.close()

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self.carl_env, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None) -> None:
        setattr(self.carl_env, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self.carl_env, method_name)(*args, **kwargs)] * n

    @staticmethod
    def _host(obs):
        if torch.is_tensor(obs):
            return obs.cpu().numpy()
        return {k: (SB3VecEnv._host(v) if not isinstance(v, dict) else {kk: vv.cpu().numpy() for kk, vv in v.items()})
                for k, v in obs.items()}
