"""Family adapter for the Brax-locomotion envs (reference: carl/envs/brax/carl_brax_env.py:115-336).

The reference creates ``brax.envs.create(env_name, backend="spring", batch_size=...)``, wraps it
with ``GymWrapper`` / ``VectorGymWrapper`` (carl/envs/brax/wrappers.py) and, on every context
change, re-parses the MJCF and rebuilds the ``System`` (``_update_context``, :255-292) -- which,
because of Quirk B1, never reaches the jitted step.  Here ``env_name`` selects a
model table of the lane engine, and the context -> physics mapping the reference INTENDS
(gravity, friction, elasticity, ang_damping, ``mass_<link>``) is applied per lane inside the
kernels; ``reference_compat=True`` ignores physics contexts, which is what the reference
effectively simulates.  ``viscosity`` is observed-only (Quirk B2).
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from carl_amd import spaces
from carl_amd.brax_engine import BraxVecEngine
from carl_amd.context.selection import AbstractSelector
from carl_amd.envs.brax import models
from carl_amd.envs.carl_env import CARLEnv
from carl_amd.utils.types import Context, Contexts

# features CARL can push into a brax System, plus anything starting with mass_
# (reference: carl_brax_env.py:258-269)
REGISTERED_CFS = [
    "friction", "ang_damping", "gravity", "viscosity", "elasticity",
    "target_distance", "target_direction", "target_radius",
    "joint_stiffness",  # extension (SURVEY.md Quirk B4): scales the spring backend's constraint_stiffness
]
GOAL_FEATURES = ("target_distance", "target_direction", "target_radius")


def check_context(context: dict[str, Any], registered_context_features: list[str]) -> None:
    """reference: carl_brax_env.py:104-112"""
    for cfname in context.keys():
        if cfname not in registered_context_features and not cfname.startswith("mass_"):
            raise RuntimeError(
                f"Context feature {cfname} can not be updated in the brax system. Only "
                f"{registered_context_features} are possible."
            )



def _precision_flag(substep_precision: str) -> bool:
    if substep_precision not in ("float64", "float32"):
        raise ValueError("substep_precision must be 'float64' or 'float32'")
    return substep_precision == "float32"


class CARLBraxEnv(CARLEnv):
    env_name: str
    backend: str = "spring"
    # features a subclass hands to the task instead of the brax system (CARLBraxPusher removes its goal
    # position from the context before the system update and sets the env's goal: carl_pusher.py:91-103)
    task_context_features: tuple = ()

    def __init__(
        self,
        env: BraxVecEngine | None = None,
        batch_size: int = 1,
        contexts: Contexts | None = None,
        obs_context_features: list[str] | None = None,
        obs_context_as_dict: bool = True,
        context_selector: AbstractSelector | type[AbstractSelector] | None = None,
        context_selector_kwargs: dict = None,
        use_language_goals: bool = False,
        *,
        device: str | torch.device | None = None,
        auto_reset: bool | None = None,
        seed: int = 0,
        lane_offset: int = 0,
        context_offset: int | None = None,
        reference_compat: bool = False,
        fin_capacity: int = 0,
        autotune: bool | None = None,
        autoreset: str = "redraw",
        mass_check: str = "warn",
        viscosity: str = "observed",
        substep_precision: str = "float64",
        **kwargs,
    ) -> None:
        """Reference parameters (carl_brax_env.py:119-131) plus the lane-engine ones.
        ``batch_size`` is the reference's name for the number of parallel envs (:164).

        ``viscosity``: "observed" (default: the feature only appears in the context observation, Quirk B2) or
        "reference" (the literal rule of carl_brax_env.py:276-279: the viscosity value is written into ``ang_damping``,
        overwriting the ``ang_damping`` context).

        ``substep_precision``: "float64" (default: the pose algebra of the pipeline substeps in float64, within north_star's
        1e-5 of the float64 restatement) or "float32" (OPT-IN: brax's own precision under JAX's default -- the reference
        creates its env without enabling x64, carl_brax_env.py:163-176 -- ~20 % faster, off the restatement by the amounts in
        profiles/r06_brax_fp32_deviation.txt; ``CARL_FLAG_BRAX_FP32``).

        ``mass_check``: what happens to ``mass_<link>`` contexts below the model's stability floor
        (``feature_tables.MASS_RATIO_FLOOR``) -- every value inside the reference's bounds (0.1, inf) constructs:
        "warn" (default): the EFFECTIVE mass is clamped at the floor per env inside the kernel (the context
        observation keeps the sampled value; ``effective_mass_context()`` returns the clamped one) and one warning
        names the features; "error": raise ``ValueError``
        (round 2's behaviour); "off": no clamp, no warning (such envs go non-finite within a few steps)."""
        if mass_check not in ("warn", "error", "off"):
            raise ValueError("mass_check must be 'warn', 'error' or 'off'")
        self._mass_check = mass_check
        goal_mode = False
        if contexts is not None and len(contexts):
            first = contexts[list(contexts.keys())[0]]
            if "target_distance" in first or "target_direction" in first:
                # the reference wraps the env with BraxWalkerGoalWrapper when goals vary (:195-223)
                vals = [(c.get("target_direction", first.get("target_direction", 0)),
                         c.get("target_distance", first.get("target_distance", 0))) for c in contexts.values()]
                if max(v[0] - vals[0][0] for v in vals) > 0.1 or max(v[1] - vals[0][1] for v in vals) > 0.1:
                    goal_mode = True
        names = list(self.get_context_features().keys())
        if env is None:
            sys_table = models.SYSTEMS[self.env_name](names, reference_compat=reference_compat)
            # goals vary across contexts -> the reference wraps the env with BraxWalkerGoalWrapper
            # (:195-223); here the wrapper's step/reset are an epilogue fused into the kernels
            sys_table.goal_mode = 1 if goal_mode else 0
            models.apply_viscosity_rule(sys_table, names, viscosity)  # "reference": viscosity overwrites ang_damping (:278-279)
            if mass_check != "off":  # per-env clamp of the effective mass ratio (carl_brax_ctx_map_t::mass_ratio_floor)
                from carl_amd.envs.brax.feature_tables import (COMBINED_FLOOR_SCALE, DEFAULT_MASS_RATIO_FLOOR,
                                                               MASS_RATIO_FLOOR)

                floors = MASS_RATIO_FLOOR.get(self.env_name, {})
                scale = COMBINED_FLOOR_SCALE.get(self.env_name, 2.0)
                cm = sys_table.ctx
                for k in range(cm.n_mass):
                    cm.mass_ratio_floor[k] = floors.get(names[cm.mass_row[k]], DEFAULT_MASS_RATIO_FLOOR)
                    cm.mass_ratio_floor_multi[k] = min(1.0, scale * cm.mass_ratio_floor[k])
            n_auto = batch_size > 1
            env = BraxVecEngine(
                sys_table, len(names),
                [[float(cf.default_value) for cf in self.get_context_features().values()]],
                batch_size,
                device="cuda" if device is None else device,
                auto_reset=n_auto if auto_reset is None else auto_reset,
                seed=seed, lane_offset=lane_offset, fin_capacity=fin_capacity, context_offset=context_offset,
                autoreset_mode=autoreset,  # "first_state" = brax's AutoResetWrapper (reference behaviour)
                pose_float32=_precision_flag(substep_precision),
            )
        self.use_language_goals = use_language_goals
        # the reference stacks BraxLanguageWrapper on the goal wrapper, i.e. only when goals vary (:216-218)
        self._language = bool(use_language_goals and goal_mode)
        self._goal_text: dict[Any, str] = {}
        self._was_reset = False
        super().__init__(
            env=env,
            contexts=contexts,
            obs_context_features=obs_context_features,
            obs_context_as_dict=obs_context_as_dict,
            context_selector=context_selector,
            context_selector_kwargs=context_selector_kwargs,
            **kwargs,
        )
        self._reference_compat = bool(reference_compat)
        self._check_mass_stability()
        # launch shape: time the launchable lane-group widths on this batch once (results do not
        # depend on the width); small batches keep the library's heuristic
        if (autotune if autotune is not None else batch_size >= 4096) and hasattr(self.env, "autotune"):
            self.env.autotune()

    def _check_mass_stability(self) -> None:
        """Link masses below the measured stability floor of the model's explicit spring integration
        (``feature_tables.MASS_RATIO_FLOOR``): the reference accepts any mass in (0.1, inf)
        (carl/envs/brax/carl_halfcheetah.py:37-57) -- there they never reach the physics (Quirk B1).  Here the
        kernel clamps the effective mass at the floor (``mass_check="warn"``, default: one warning), or the
        constructor refuses (``"error"``).  Not applied with ``reference_compat=True``."""
        if getattr(self, "_reference_compat", False) or getattr(self, "_mass_check", "warn") == "off":
            return
        import warnings

        from carl_amd.envs.brax.feature_tables import DEFAULT_MASS_RATIO_FLOOR, MASS_RATIO_FLOOR

        floors = MASS_RATIO_FLOOR.get(self.env_name, {})
        feats = self.get_context_features()
        names = list(self._table.names)
        low = []
        for j, name in enumerate(names):
            if not name.startswith("mass_") or name not in feats:
                continue
            col = self._table.tensor[j] if hasattr(self._table, "tensor") else self._table.values_2d[:, j]
            lowest = float(col.min())
            floor = floors.get(name, DEFAULT_MASS_RATIO_FLOOR) * float(feats[name].default_value)
            if lowest < floor:
                low.append(f"{name} = {lowest:g} < {floor:g}")
        self._check_stiffness_stability(names)
        if not low:
            return
        msg = (f"{type(self).__name__}: context values below the smallest effective mass for which this model's explicit "
               f"spring integration stays stable (measured: tools/mass_stability_sweep.py): {'; '.join(low)}")
        if self._mass_check == "error":
            raise ValueError(msg + "; pass mass_check='warn' to clamp the effective mass per env, or reference_compat=True "
                             "to ignore physics contexts as the reference effectively does")
        warnings.warn(msg + " -- the physics runs these envs at the floor (the context observation keeps the sampled "
                      "value); mass_check='error' refuses instead", RuntimeWarning, stacklevel=3)

    def _check_stiffness_stability(self, names) -> None:
        """``joint_stiffness`` (extension feature of the ``...Stiffness`` classes; declared bounds (0.01, 100)) above the measured
        ceiling of the model (``feature_tables.JOINT_STIFFNESS_CEILING``): every env there leaves the finite range within a few
        hundred steps.  The physics is not clamped (a stiffer joint is what was asked for): one RuntimeWarning with
        ``mass_check="warn"``, a ValueError with ``"error"``."""
        import warnings

        from carl_amd.envs.brax.feature_tables import JOINT_STIFFNESS_CEILING

        ceiling = JOINT_STIFFNESS_CEILING.get(self.env_name)
        if ceiling is None or "joint_stiffness" not in names:
            return
        j = names.index("joint_stiffness")
        col = self._table.tensor[j] if hasattr(self._table, "tensor") else self._table.values_2d[:, j]
        highest = float(col.max())
        if highest <= ceiling:
            return
        msg = (f"{type(self).__name__}: joint_stiffness = {highest:g} > {ceiling:g}, the largest scale of the constraint stiffness "
               f"for which this model's explicit spring integration stays stable at default masses (measured: "
               f"tools/stiffness_stability_sweep.py)")
        if getattr(self, "_mass_check", "warn") == "error":
            raise ValueError(msg + "; pass mass_check='warn' to run such envs anyway (they leave the finite range)")
        warnings.warn(msg + " -- envs above it leave the finite range within a few hundred steps (the physics is not clamped); "
                      "mass_check='error' refuses instead", RuntimeWarning, stacklevel=4)

    def effective_mass_context(self) -> dict[str, torch.Tensor]:
        """The ``mass_<link>`` context values the PHYSICS runs each env at: ``{feature: [N] float32}`` in the
        feature's own unit (kg).  Equal to the sampled context -- what ``obs["context"]`` reports -- except where
        ``mass_check="warn"`` clamps the effective mass ratio at the model's stability floor
        (``carl_brax_ctx_map_t::mass_ratio_floor`` / ``_multi``): a context-conditioned policy that wants the mass
        the dynamics actually use reads it here.  Same rule as ``load_ctx`` in
        carl_amd/csrc/brax_kernels.hip.h: ratio = value / CARL default; the higher ``_multi`` floor applies to an
        env in which two or more links are lighter than nominal (ratio < 0.999)."""
        eng = self.env
        cm = eng.sys.ctx
        table = eng.ctx_table  # [F][C] float32 on the device
        idx = eng.ctx_idx.long()
        names = list(self.get_context_features().keys())
        out: dict[str, torch.Tensor] = {}
        if cm.n_mass == 0:
            return out
        ratios = [table[cm.mass_row[k]][idx] / float(cm.mass_nominal[k]) for k in range(cm.n_mass)]
        n_light = sum((r < 0.999).to(torch.int32) for r in ratios)
        compat = getattr(self, "_reference_compat", False)
        for k, r in enumerate(ratios):
            floor = torch.where(n_light >= 2, torch.full_like(r, float(cm.mass_ratio_floor_multi[k])),
                                torch.full_like(r, float(cm.mass_ratio_floor[k])))
            eff = r if compat else torch.maximum(r, floor)
            out[names[cm.mass_row[k]]] = eff * float(cm.mass_nominal[k])
        return out

    def _base_observation_space(self) -> spaces.Space:
        obs = np.inf * np.ones(self.env.D, dtype=np.float32)
        return spaces.Box(-obs, obs, dtype=np.float32)  # wrappers.py:46-47

    def _action_space(self) -> spaces.Space:
        s = self.env.sys
        lo = np.array(s.act_lo[: s.n_act], dtype=np.float32)
        hi = np.array(s.act_hi[: s.n_act], dtype=np.float32)
        return spaces.Box(lo, hi, dtype=np.float32)  # wrappers.py:50-51 (sys.actuator.ctrl_range)

    def _update_context(self) -> None:
        # task features are taken out before the system check, as CARLBraxPusher._update_context does
        # with its goal position (carl_pusher.py:91-103)
        check_context(self.context, REGISTERED_CFS + list(self.task_context_features))
        super()._update_context()

    @property
    def contexts(self) -> Contexts:
        return self._contexts

    @contexts.setter
    def contexts(self, contexts: Contexts) -> None:
        for c in contexts.values() if not hasattr(contexts, "names") else []:
            check_context(c, REGISTERED_CFS + list(self.task_context_features))
        CARLEnv.contexts.fset(self, contexts)
        if hasattr(self, "_reference_compat"):
            self._check_mass_stability()

    # ---- language goals (carl/envs/brax/brax_walker_goal_wrapper.py:143-181) -------------------------
    @staticmethod
    def describe_goal(context: Context) -> str:
        """the sentence ``BraxLanguageWrapper.get_goal_desc`` builds from a context's goal features"""
        from carl_amd.envs.brax.brax_walker_goal_wrapper import DIRECTION_NAMES

        where = f"{context['target_distance']}m {DIRECTION_NAMES[context['target_direction']]}"
        if "target_radius" in context:
            return f"The distance to the goal is {where}. Move within {context['target_radius']} steps of the goal."
        return f"Move {where}."

    def _with_goal_text(self, obs: dict, context_ids) -> dict:
        """``obs["obs"]`` becomes ``{"obs": ..., "goal": text}`` -- one string for the scalar API, a list with
        one string per env for a batch (built from the envs' context ids on the host: strings do not live
        in HBM; texts are cached per context)"""
        if not self._language:
            return obs
        if self._scalar_api:
            text: Any = self.describe_goal(self.context)
        else:
            keys = self.context_selector.contexts_keys
            text = []
            for cid in (context_ids.tolist() if torch.is_tensor(context_ids) else context_ids):
                if cid not in self._goal_text:
                    self._goal_text[cid] = self.describe_goal(self.contexts[keys[cid]])
                text.append(self._goal_text[cid])
        obs["obs"] = {"obs": obs["obs"], "goal": text}
        return obs

    @property
    def position(self):
        """the goal wrapper's integrated planar position: ``None`` until the first reset (and without goal
        mode), then (x, y) -- a length-2 array for one env, an ``[N, 2]`` tensor for a batch"""
        if not self.env.sys.goal_mode or not self._was_reset:
            return None
        pos = self.env.goal_pos  # [2][N]
        return pos[:, 0].cpu().numpy() if self._scalar_api else pos.t()

    def step(self, action: Any):
        if self._scalar_api:  # one env: action is a length-A vector
            self.env.step(self.env.stage_scalar_action(np.asarray(action, dtype=np.float32)))
            o, r, te, tr = self.env.read_transition()  # one device-to-host copy for the whole transition
            state = o[0].astype(np.float32)
            info: dict[str, Any] = {"context_id": self.context_id}
            # wrappers.py:76-77: terminated = done, truncated = False; brax's EpisodeWrapper
            # folds its 1000-step truncation into `done`
            done = bool(te[0]) or bool(tr[0])
            if self.env.sys.goal_mode:
                info["success"] = int(self.env.success[0])
            return self._with_goal_text(self._add_context_to_state(state), None), float(r[0]), done, False, info
        out = super().step(action)
        if self.env.sys.goal_mode:
            out[4]["success"] = self.env.success
        self._with_goal_text(out[0], out[4].get("context_id"))
        return out

    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None):
        obs, info = super().reset(seed=seed, options=options)
        self._was_reset = True
        if self.env.sys.goal_mode:  # brax_walker_goal_wrapper.py:121
            info["success"] = 0 if self._scalar_api else torch.zeros_like(self.env.success)
        return self._with_goal_text(obs, info.get("context_id")), info

    @classmethod
    def get_default_context(cls) -> Context:
        """Default context without goal features (reference: :308-324)."""
        default_context = cls.get_context_space().get_default_context()
        for k in GOAL_FEATURES:
            default_context.pop(k, None)
        return default_context

    @classmethod
    def get_default_goal_context(cls) -> Context:
        return cls.get_context_space().get_default_context()
