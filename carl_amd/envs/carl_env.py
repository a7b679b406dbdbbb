"""CARLEnv -- the contextual reset/step/context API of the reference
(carl/envs/carl_env.py:19-342), re-provided on top of the MI355X lane engine.

Two modes, chosen by ``num_envs``:

* ``num_envs == 1`` (default) -- the reference's scalar API, value for value:
  ``reset() -> ({"obs": np.float32[D], "context": {...}}, {"context_id": int})``,
  ``step(a) -> (obs, float, bool, bool, info)``; the host selector object drives the
  context (any selector class, ``n_calls`` / ``context_id`` semantics identical); no
  auto-reset (the gymnasium path has none).  The physics still runs on the GPU lane
  engine -- there is no CPU path.
* ``num_envs == N > 1`` -- N lanes in HBM: observations, rewards and flags are
  device tensors ``[N, ...]`` that ALIAS the engine's output buffers (no per-step allocation: the next ``step``
  overwrites them -- ``clone()`` what must outlive it; gymnasium's VectorEnv returns copies); ``info["context_id"]`` is an int32 tensor; done lanes are
  reset inside ``step`` (``auto_reset=True``; returned obs = reset obs, terminal obs in
  ``info["final_observation"]``); each lane advances its own selector state on device
  (static / round robin / random rules of carl/context/selection.py).
"""
from __future__ import annotations

import abc
import inspect
from typing import Any

import numpy as np
import torch

from carl_amd import _lib, spaces
from carl_amd.context.context_space import ContextFeature, ContextSpace
from carl_amd.context.selection import (
    SEL_HOST,
    AbstractSelector,
    RoundRobinSelector,
)
from carl_amd.context.table import ContextTable
from carl_amd.engine import VecEngine
from carl_amd.utils.types import Context, Contexts


class _Unwrapped:
    """Stand-in for ``env.unwrapped`` of the gymnasium env: attribute reads return the
    parameter of lane 0's current context, attribute writes broadcast a scalar into the
    whole context column (the setattr protocol of carl_gymnasium_env.py:75-77);
    ``state`` reads/writes lane state (the reference's reset overrides assign
    ``env.unwrapped.state``, e.g. carl_cartpole.py:51)."""

    def __init__(self, owner: "CARLEnv"):
        object.__setattr__(self, "_owner", owner)

    def __getattr__(self, name):
        owner = object.__getattribute__(self, "_owner")
        eng = owner.env
        if name == "state":
            s = eng.state.t()
            return s[0].cpu().numpy().astype(np.float64) if eng.n == 1 else s
        names = owner._feature_names
        if name in names:
            col = eng.ctx_table[names.index(name)]
            return float(col[int(eng.ctx_idx[0])])
        raise AttributeError(name)

    def __setattr__(self, name, value):
        owner = object.__getattribute__(self, "_owner")
        eng = owner.env
        if name == "state":
            v = torch.as_tensor(np.asarray(value, dtype=np.float32)).reshape(-1, eng.S)
            eng.state.copy_(v.t().to(eng.device).expand(eng.S, eng.n) if v.shape[0] == 1 else v.t().to(eng.device))
            return
        names = owner._feature_names
        if name in names:
            eng.ctx_table[names.index(name)].fill_(float(value))
            if eng.n > 1:  # (scalar mode builds its context observation on the host)
                eng.refresh_ctx_obs()
            return
        object.__setattr__(self, name, value)


class CARLEnv(abc.ABC):
    metadata: dict = {}

    def __init__(
        self,
        env: VecEngine | None = None,
        contexts: Contexts | None = None,
        obs_context_features: list[str] | None = None,
        obs_context_as_dict: bool = True,
        context_selector: AbstractSelector | type[AbstractSelector] | None = None,
        context_selector_kwargs: dict | None = None,
        **kwargs,
    ):
        """Same parameters as the reference's ``CARLEnv.__init__`` (carl_env.py:20-29).
        ``env`` is the lane engine (built by the family adapter when None)."""
        if env is None:
            raise ValueError("CARLEnv needs a lane engine; use a family class such as CARLCartPole")
        self.env = env
        self.num_envs = env.n
        self._scalar_api = bool(kwargs.pop("_scalar_api", env.n == 1))
        self._feature_names = list(self.get_context_features().keys())
        self.obs_context_as_dict = obs_context_as_dict

        if contexts is None:
            contexts = {0: self.get_default_context()}
        self.contexts = contexts  # setter: fills defaults, builds the dense table
        self.context: Context | None = None
        if obs_context_features is None:
            obs_context_features = list(self._table.names)
        self.obs_context_features = obs_context_features

        # Context selector (reference: carl_env.py:91-108)
        if context_selector is None:
            self.context_selector = RoundRobinSelector(contexts=self._contexts)
        elif isinstance(context_selector, AbstractSelector):
            self.context_selector = context_selector
        elif inspect.isclass(context_selector) and issubclass(context_selector, AbstractSelector):
            if context_selector_kwargs is None:
                context_selector_kwargs = {}
            context_selector_kwargs.update({"contexts": self._contexts})
            self.context_selector = context_selector(**context_selector_kwargs)
        else:
            raise ValueError(
                f"Context selector must be None or an AbstractSelector class or instance. "
                f"Got type {type(context_selector)}."
            )
        self._configure_engine()

        self.base_observation_space = self._base_observation_space()
        self.single_observation_space = self.get_observation_space(
            obs_context_feature_names=self.obs_context_features
        )
        self.single_action_space = self._action_space()
        if self._scalar_api:
            self.observation_space = self.single_observation_space
            self.action_space = self.single_action_space
        else:
            self.observation_space = spaces.batch_space(self.single_observation_space, self.num_envs)
            self.action_space = spaces.batch_space(self.single_action_space, self.num_envs)
        self.unwrapped_env = _Unwrapped(self)

    # ------------------------------------------------------------------ engine glue
    def _configure_engine(self) -> None:
        eng = self.env
        rule = getattr(self.context_selector, "device_rule", SEL_HOST)
        if self._scalar_api:
            rule = SEL_HOST  # the host selector object decides, exactly like the reference
        elif rule == SEL_HOST:
            raise ValueError(
                f"{type(self.context_selector).__name__} runs on the host and cannot drive "
                f"{self.num_envs} auto-resetting lanes; use Static/RoundRobin/RandomSelector or num_envs=1"
            )
        eng.b.selector = int(rule)
        eng.b.selector_stride = int(getattr(self.context_selector, "stride", 1))
        rows = [self._table.names.index(n) for n in self.obs_context_features
                if n in self._table.names and self._table.names.index(n) < eng.F]
        eng.ctx_obs_rows = rows
        eng.ctx_obs = torch.zeros((len(rows), eng.n), dtype=torch.float32, device=eng.device)
        if hasattr(self._table, "tensor"):  # DeviceContextTable: already [F][C] in HBM
            eng.set_contexts_device(self._table.tensor[: eng.F])
        else:
            eng.set_contexts(self._table.values_2d[:, : eng.F])
        eng._sync_pointers()

    # ------------------------------------------------------------------ properties
    @property
    def contexts(self) -> Contexts:
        return self._contexts

    @contexts.setter
    def contexts(self, contexts: Contexts) -> None:
        """Fill every context with defaults (reference: carl_env.py:122-137) and keep
        the dense form the engine uploads."""
        space = self.get_context_space()
        if hasattr(contexts, "tensor"):  # DeviceContextTable (carl_amd/context/device_sampler.py)
            if list(contexts.names) != list(space.context_feature_names):
                raise ValueError("a device context table must hold the env's context features in table order")
            self._table = contexts
            self._contexts = contexts
        elif isinstance(contexts, ContextTable):
            self._table = space.to_table(contexts)
            self._contexts = self._table
        else:
            self._contexts = {k: space.insert_defaults(v) for k, v in contexts.items()}
            self._table = ContextTable.from_contexts(
                self._contexts, self._union_names(self._contexts, space), self._defaults_for(self._contexts, space))
        if getattr(self, "context_selector", None) is not None and hasattr(self, "obs_context_features"):
            self.context_selector.contexts = self._contexts
            self.context_selector.context_ids = list(np.arange(len(self._contexts)))
            self.context_selector.contexts_keys = list(self._contexts.keys())
            self._configure_engine()

    @staticmethod
    def _union_names(contexts, space: ContextSpace) -> list[str]:
        names = list(space.context_feature_names)
        for c in contexts.values():
            for k in c:
                if k not in names:
                    names.append(k)
        return names

    @staticmethod
    def _defaults_for(contexts, space: ContextSpace) -> dict:
        d = dict(space.get_default_context())
        for c in contexts.values():
            for k, v in c.items():
                d.setdefault(k, v)
        return d

    @property
    def context_id(self):
        if self._scalar_api:
            return self.context_selector.context_id
        return self.env.ctx_idx

    @context_id.setter
    def context_id(self, new_id) -> None:
        """Switch the context immediately (reference: carl_env.py:139-157)."""
        assert new_id in self.context_selector.context_ids, (
            "Unknown ID, this context does not exist in the context set."
        )
        self.context_selector.context_id = new_id
        self.context_selector.context = self.context_selector.contexts[
            self.context_selector.contexts_keys[new_id]]
        self.context = self.context_selector.context
        self._update_context()

    @property
    def unwrapped(self):
        return self.unwrapped_env

    # ------------------------------------------------------------------ spaces
    def get_observation_space(self, obs_context_feature_names: list[str] | None = None) -> spaces.Dict:
        context_space = self.get_context_space()
        obs_space_context = context_space.to_gymnasium_space(
            context_feature_names=obs_context_feature_names, as_dict=self.obs_context_as_dict)
        return spaces.Dict({"obs": self.base_observation_space, "context": obs_space_context})

    @abc.abstractmethod
    def _base_observation_space(self) -> spaces.Space:
        ...

    @abc.abstractmethod
    def _action_space(self) -> spaces.Space:
        ...

    @staticmethod
    @abc.abstractmethod
    def get_context_features() -> dict[str, ContextFeature]:
        ...

    @classmethod
    def get_context_space(cls) -> ContextSpace:
        return ContextSpace(cls.get_context_features())

    @classmethod
    def get_default_context(cls) -> Context:
        return cls.get_context_space().get_default_context()

    # ------------------------------------------------------------------ context handling
    def _progress_instance(self) -> None:
        """Select the next context with the (host) selector (carl_env.py:228-243)."""
        self.context = self.context_selector.select()

    def _update_context(self) -> None:
        """Point lane(s) at the selected context.  The table is already resident, so
        what the reference does with a setattr loop (carl_gymnasium_env.py:75-77) is one
        int32 write here."""
        cid = self.context_selector.context_id
        if cid is None:
            raise RuntimeError("`_progress_instance` must be called before `_update_context`")
        self.env.ctx_idx.fill_(int(cid))
        if not self._scalar_api:  # the kernels rewrite a lane's context observation only when a reset moves it
            self.env.refresh_ctx_obs()

    # ------------------------------------------------------------------ reset / step
    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None):
        """carl_env.py:245-274.  Scalar mode: host selector -> ``_update_context`` only if
        the id changed -> engine reset.  Batched mode: every lane advances its own selector
        state and draws its init state on device."""
        if seed is not None:
            self.env.seed(seed)
        if self._scalar_api:
            last_context_id = self.context_id
            self._progress_instance()
            if self.context_id != last_context_id:
                self._update_context()
            obs = self.env.reset()
            state = obs[0].cpu().numpy()
            info: dict[str, Any] = {"context_id": self.context_id}
            return self._add_context_to_state(state), info
        self.env.reset()
        # host selector bookkeeping mirrors the number of reset() calls
        self.context_selector.n_calls += 1
        return self._batched_obs(), {"context_id": self.env.ctx_idx}

    def step(self, action: Any):
        """carl_env.py:321-342."""
        if self._scalar_api:
            self.env.step(self.env.stage_scalar_action(action))
            o, r, te, tr = self.env.read_transition()  # one device-to-host copy for the whole transition
            state = o[0].astype(np.float32)
            info: dict[str, Any] = {"context_id": self.context_id}
            return self._add_context_to_state(state), float(r[0]), bool(te[0] != 0), bool(tr[0] != 0), info
        # Batched hot call: everything returned is a view of an engine buffer whose address never changes, so
        # the views (and the observation dict) are built once; per call this is the engine's launch plus one
        # small dict.  `_final_observation` is the done mask the step kernel writes (carl_step_io_t::done).
        c = self._views()
        _, reward, _, _ = self.env.step(action)
        info = dict(c["info_auto"] if self.env.auto_reset else c["info_plain"])
        # the observation dict (and its nested context dict) is a fresh shallow copy per call, like `info` and
        # like reset(): a caller that replaces or pops an entry does not corrupt later returns; the tensors inside
        # are the cached views
        ctx = c["obs"]["context"]
        obs = {"obs": c["obs"]["obs"], "context": dict(ctx) if isinstance(ctx, dict) else ctx}
        return obs, reward, c["term"], c["trunc"], info

    def _views(self) -> dict:
        eng = self.env
        key = (eng.obs.data_ptr(), eng.ctx_obs.data_ptr(), eng.ctx_idx.data_ptr(), eng.terminated.data_ptr(),
               eng.done.data_ptr(), tuple(eng.ctx_obs_rows), self.obs_context_as_dict)
        c = getattr(self, "_view_cache", None)
        if c is None or c["key"] != key:  # buffers are re-homed by contexts= / MixedVecEngine: rebuild then
            info_plain = {"context_id": eng.ctx_idx}
            c = {"key": key, "obs": self._batched_obs(), "term": eng.terminated.view(torch.bool),
                 "trunc": eng.truncated.view(torch.bool), "info_plain": info_plain,
                 "info_auto": {**info_plain, "final_observation": eng.final_obs,
                               "_final_observation": eng.done.view(torch.bool)}}
            self._view_cache = c
        return c

    def _batched_obs(self) -> dict[str, Any]:
        eng = self.env
        if self.obs_context_as_dict:
            ctx = {self._table.names[r]: eng.ctx_obs[k] for k, r in enumerate(eng.ctx_obs_rows)}
        else:
            ctx = eng.ctx_obs.t()
        return {"obs": eng.obs, "context": ctx}

    def _add_context_to_state(self, state: Any) -> dict[str, Any]:
        """carl_env.py:276-305 (Quirk S5: dict mode keeps context order, vector mode
        keeps ``obs_context_features`` order)."""
        if not self.obs_context_as_dict:
            context = [self.context[k] for k in self.obs_context_features]
        else:
            context = {k: v for k, v in self.context.items() if k in self.obs_context_features}
        return {"obs": state, "context": context}

    def close(self) -> None:
        pass

    def render(self):
        raise NotImplementedError("rendering is outside the engine's scope")
