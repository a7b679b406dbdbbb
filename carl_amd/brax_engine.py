"""Lane engine for the Brax-locomotion families: ``VecEngine`` with a model table
(``carl_brax_sys_t``) instead of a built-in family.  Replaces, for N lanes at once, what
``CARLBraxEnv`` builds with ``brax.envs.create(env_name, backend="spring", batch_size)`` +
``VectorGymWrapper`` (reference: carl/envs/brax/carl_brax_env.py:163-190,
carl/envs/brax/wrappers.py:93-158).  No CPU path."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from carl_amd import _lib
from carl_amd.engine import VecEngine, _ptr


@dataclass
class _BraxInfo:
    state_dim: int
    obs_dim: int
    n_features: int
    action_dim: int
    action_is_discrete: int
    n_actions: int
    max_episode_steps: int
    action_low: float
    action_high: float


class BraxVecEngine(VecEngine):
    _narrow_actions = False  # carl_brax_step / carl_brax_rollout read float32 actions: narrower dtypes are widened here

    def __init__(self, sys_table: _lib.BraxSys, n_features: int, ctx_table, n_lanes: int, device="cuda", *,
                 autoreset_mode: str = "redraw", **kw):
        """``autoreset_mode``: what the in-kernel auto-reset does with a done env --
        "redraw" (default): selector advance, new init-state draw, like an explicit reset;
        "first_state": put the env back to the state its last explicit ``reset()`` produced, same context,
        nothing drawn -- brax's ``AutoResetWrapper`` as the reference reaches it (wrappers.py:54-78,121-145)."""
        if autoreset_mode not in ("redraw", "first_state"):
            raise ValueError("autoreset_mode must be 'redraw' or 'first_state'")
        self.sys = sys_table
        self._n_features = int(n_features)
        kw.pop("cartpole_recompute", None)
        self.goal_pos = self.success = self.first_state = self.branch_sig = None
        self.autoreset_mode = autoreset_mode
        branch_record = bool(kw.pop("branch_record", False))
        generic = bool(kw.pop("generic_substep", False))  # planar models: step with the general 3-D substep (A/B, tests)
        # OPT-IN, never a default: the substeps' pose algebra in float32 -- brax's own precision under JAX's default; faster,
        # and off the float64 restatement by more than north_star's 1e-5 (include/carl_amd.h: CARL_FLAG_BRAX_FP32)
        pose_float32 = bool(kw.pop("pose_float32", False))
        if pose_float32 and (sys_table.target_link > 0 or sys_table.push_link > 0):
            raise ValueError("pose_float32 is not built for the reach / push task models")
        super().__init__(-1, ctx_table, n_lanes, device, **kw)
        if generic:
            self.b.flags |= _lib.FLAG_BRAX_GENERIC
        if pose_float32:
            self.b.flags |= _lib.FLAG_BRAX_FP32
        if autoreset_mode == "first_state":
            self.b.flags |= _lib.FLAG_AUTORESET_FIRST_STATE
            self.first_state = torch.zeros((self.n, self.S), dtype=torch.float32, device=self.device)
        # Brax state is env-major in HBM ([N][L][20]: per link pose head 7 | pose tail 7 | velocities 6,
        # include/carl_amd.h ABI 8); ``self.state`` is the [20 L, N] VIEW of that storage (the classic-control
        # engine's indexing); ``state64()`` returns it as float64 [N, L, 13].
        self._state_storage = torch.zeros((self.n, self.S), dtype=torch.float32, device=self.device)
        self.state = self._state_storage.t()
        if branch_record:  # per-step hash of the physics' discrete decisions (carl_step_io_t::branch_sig)
            self.branch_sig = torch.zeros((self.n, 2), dtype=torch.int32, device=self.device)
        self._sync_pointers()
        if self.sys.goal_mode:  # BraxWalkerGoalWrapper state: integrated (x, y) + per-step success flag
            self.goal_pos = torch.zeros((2, self.n), dtype=torch.float32, device=self.device)
            self.success = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            self._sync_pointers()
        # device copy of the model table (the kernels stage it into LDS once per workgroup)
        raw = bytes(self.sys)
        self.sys_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self._sys_ref = C.byref(self.sys)

    def _family_info(self):
        s = self.sys
        return _BraxInfo(_lib.BRAX_LINK_RECORD * s.n_links, s.obs_dim, self._n_features, s.n_act, 0, 0,
                         s.max_episode_steps, float(min(s.act_lo[: s.n_act])), float(max(s.act_hi[: s.n_act])))

    def _c_reset(self, mask_ptr) -> int:
        return self.lib.carl_brax_reset(C.byref(self.b), _ptr(self.sys_dev), C.byref(self.sys), mask_ptr,
                                        _ptr(self.obs), self._stream())

    def _c_step(self, io) -> int:
        return self.lib.carl_brax_step(C.byref(self.b), _ptr(self.sys_dev), C.byref(self.sys), C.byref(io),
                                       self._stream())

    def _c_step_fast(self, stream: int) -> int:
        return self.lib.carl_brax_step(self._b_ref, self.sys_dev.data_ptr(), self._sys_ref, self._io_ref, stream)

    def _c_rollout(self, io, n_steps: int) -> int:
        return self.lib.carl_brax_rollout(C.byref(self.b), _ptr(self.sys_dev), C.byref(self.sys), C.byref(io),
                                          n_steps, self._stream())

    # ------------------------------------------------------------------ state access
    def state64(self) -> torch.Tensor:
        """The envs' maximal-coordinate state as float64 ``[N, L, 13]`` (per link COM position 3, rotation 4
        (w, x, y, z), linear velocity 3, angular velocity 3): pose = head + tail of the HBM record."""
        L = self.sys.n_links
        rec = self._state_storage.to(torch.float64).reshape(self.n, L, _lib.BRAX_LINK_RECORD)
        return torch.cat([rec[:, :, :7] + rec[:, :, 7:14], rec[:, :, 14:]], dim=2)

    def state_np(self):
        """``state64()`` on the host as ``[N, 13 L]`` float64 (the oracle's layout)"""
        return self.state64().reshape(self.n, 13 * self.sys.n_links).cpu().numpy()

    def set_state64(self, state) -> None:
        """Inverse of ``state64``: ``[N, L, 13]`` (or ``[N, 13 L]``) float64 -> the HBM record (pose split into a
        float32 head and tail; velocities rounded to float32).  Planar models (Halfcheetah, Hopper, Walker2d) are
        stepped by a substep that assumes the state lies in the y = 0 plane -- what ``reset`` produces; a state
        that does not switches the engine to the general substep (``_check_planar_state``).  Direct writes to
        ``eng.state`` bypass that check: pass ``generic_substep=True`` if you write out-of-plane states yourself."""
        L = self.sys.n_links
        st = torch.as_tensor(state, dtype=torch.float64, device=self.device).reshape(self.n, L, 13)
        self._check_planar_state(st)
        # (rotations are taken as given.  The engine's own reset makes its rotations unit quaternions in float64 -- round 6 --
        # because rotation formulas that agree for unit quaternions differ by (|q|^2 - 1) x the vector otherwise; a caller
        # who wants the float64 restatement's numbers to 1e-5 on the first substep passes unit quaternions too.)
        pose = st[:, :, :7]
        head = pose.to(torch.float32)
        tail = (pose - head.to(torch.float64)).to(torch.float32)
        self._state_storage.copy_(torch.cat([head, tail, st[:, :, 7:].to(torch.float32)], dim=2).reshape(self.n, -1))

    def is_planar_model(self) -> bool:
        """True when the library steps this model with the planar substep (Halfcheetah, Hopper, Walker2d: every joint
        axis is +-y and all geometry lies in the y = 0 plane) -- unless ``generic_substep=True`` was passed."""
        return bool(self.lib.carl_brax_model_is_planar(C.byref(self.sys))) and not (self.b.flags & _lib.FLAG_BRAX_GENERIC)

    def _check_planar_state(self, st: torch.Tensor) -> None:
        """The planar substep neither reads nor updates the out-of-plane components of a state (y positions, the x / z
        quaternion components, v_y, w_x, w_z): a caller-provided state that is not planar would silently get wrong
        physics.  Such a state switches this engine to the general 3-D substep, with one warning."""
        if not self.is_planar_model():
            return
        off = torch.stack([st[:, :, 1].abs().amax(), st[:, :, 4].abs().amax(), st[:, :, 6].abs().amax(),
                           st[:, :, 8].abs().amax(), st[:, :, 10].abs().amax(), st[:, :, 12].abs().amax()]).amax()
        if float(off) > 1e-4:  # (reset's float32 kinematics leaves <= 1e-5 out of plane)
            import warnings

            self.b.flags |= _lib.FLAG_BRAX_GENERIC
            warnings.warn(f"set_state64: the state leaves the y = 0 plane of a planar model (largest out-of-plane "
                          f"component {float(off):.3g}); this engine now steps it with the general 3-D substep "
                          "(CARL_FLAG_BRAX_GENERIC)", RuntimeWarning, stacklevel=3)

    def rollout_variant(self) -> int:
        """Brax families have ONE rollout kernel (no staged / direct-store pair): nothing to warn about."""
        return _lib.ROLLOUT_STAGED

    _has_direct_kernel = False

    def _row_pitch(self) -> int:
        return self.n  # dense rows: the Brax kernel writes per-env pieces, not 16-byte pieces of lane rows

    def alloc_rollout(self, n_steps: int, final_obs: bool = False, branch_record: bool = False) -> dict:
        out = super().alloc_rollout(n_steps, final_obs)
        if branch_record:
            out["branch_sig"] = torch.zeros((n_steps, self.n, 2), dtype=torch.int32, device=self.device)
        if self.sys.goal_mode:
            out["success"] = torch.zeros((n_steps, self.n), dtype=torch.uint8, device=self.device)
        return out

    def _shape_for(self, n_steps: int) -> None:
        """Once ``autotune`` has been used on this engine, launches keep to the lane-group width that was fastest for THEIR
        length class: launches of >= 4 env steps run the balanced fragment schedule, shorter ones one wavefront per group
        (carl_brax.hip: launch_brax), and the best width differs (Halfcheetah x 32 768: a 2-step probe picks 8 lanes per
        env, 50-step rollouts are 30 % faster at 7).  The class not probed yet is probed on first use (state saved and
        restored; results never depend on the width)."""
        tuned = getattr(self, "_tuned", None)
        if tuned is None or getattr(self, "_tuning", False) or torch.cuda.is_current_stream_capturing():
            return  # (never probe inside a hipGraph capture: the width in force is recorded as it is)
        long_launch = n_steps >= 4
        if long_launch not in tuned:  # only after autotune(both_classes=False)
            self._probe_widths(8 if long_launch else 2, 2)
        self.sys.lanes_per_env = tuned[long_launch]

    def step(self, action):
        self._shape_for(1)
        return super().step(action)

    def rollout(self, actions, out: dict | None = None) -> dict:
        self._shape_for(int(actions.shape[0]))
        if not self.sys.goal_mode:
            return super().rollout(actions, out)
        if out is None:
            out = self.alloc_rollout(int(actions.shape[0]))
        self.b.success = out["success"].data_ptr()  # [T][N] for the duration of this launch
        try:
            return super().rollout(actions, out)
        finally:
            self.b.success = self.success.data_ptr()

    # ------------------------------------------------------------------ launch-shape autotuning
    def lane_widths(self) -> list[int]:
        """Lane-group widths (lanes sharing one env) the library can launch for this model."""
        out = (C.c_int32 * 16)()
        n = self.lib.carl_brax_lane_widths(C.byref(self.sys), int(self.b.flags), out, 16)
        return [int(out[i]) for i in range(n)]

    _PROBE_SAVED = ("state", "elapsed", "ctx_idx", "episode", "n_calls", "ep_return", "last_return", "last_length",
                    "episodes_done", "obs", "ctx_obs", "reward", "terminated", "truncated", "done", "final_obs",
                    "goal_pos", "success", "fin_count", "first_state", "branch_sig")

    def autotune(self, n_steps: int = 2, reps: int = 2, both_classes: bool = True) -> int:
        """Time ``carl_brax_rollout`` on THIS batch for every launchable lane-group width and keep
        the fastest (``sys.lanes_per_env``).  The width is a pure scheduling choice -- results are
        bit-identical across widths (tests/test_gpu_brax.py) -- but the best one depends on the
        batch size AND on the launch length: what matters is how evenly ceil(N / envs_per_wave) groups
        fill the resident wave slots of the chip, the instruction count per wave, and -- for batches
        larger than the chip holds at once -- how the kernel's (group, step-range) fragments divide
        (Halfcheetah x 32 768: width 4 wins a 2-step probe, width 7 is 10 % faster at 20 steps:
        ``tools/autotune_probe.py``).  Pass the ``n_steps`` the launches will have (per-call stepping:
        the default); with ``both_classes`` (default) the OTHER launch-length class (>= 4 env steps / shorter) is
        probed here too, so that ``step`` / ``rollout`` never stop to probe -- a probe synchronises the device.
        Every buffer a launch can change is saved and restored around the probe (also when a probe launch raises);
        the probe's actions come from a private generator (the caller's CUDA RNG stream is not advanced)."""
        best = self._probe_widths(n_steps, reps)
        if both_classes:
            other = not (n_steps >= 4)
            if other not in self._tuned:
                self._probe_widths(8 if other else 2, reps)
            self.sys.lanes_per_env = best
        return best

    def _probe_widths(self, n_steps: int, reps: int) -> int:
        saved = {k: getattr(self, k).clone() for k in self._PROBE_SAVED
                 if isinstance(getattr(self, k, None), torch.Tensor)}
        width0, tuning0 = int(self.sys.lanes_per_env), getattr(self, "_tuning", False)
        lo, hi = float(min(self.sys.act_lo[: self.sys.n_act])), float(max(self.sys.act_hi[: self.sys.n_act]))
        gen = torch.Generator(device=self.device).manual_seed(0x5EED)
        acts = torch.rand((n_steps, self.n, self.sys.n_act), device=self.device, generator=gen) * (hi - lo) + lo
        out = self.alloc_rollout(n_steps)
        best, best_ms = 0, float("inf")
        timings = {}
        self._tuning = True  # (the probe's own launches keep the width under test)
        try:
            if int(self.episode.max()) == 0:  # never reset: give the probe a valid state
                self.reset()
            for w in self.lane_widths():
                self.sys.lanes_per_env = w
                self.rollout(acts, out)  # warm-up (code object load)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    self.rollout(acts, out)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / reps
                timings[w] = ms
                if ms < best_ms:
                    best, best_ms = w, ms
        except BaseException:
            self.sys.lanes_per_env = width0  # (a failed probe changes nothing)
            raise
        else:
            self.sys.lanes_per_env = best
            self.autotune_ms = timings
            if getattr(self, "_tuned", None) is None:
                self._tuned = {}
            self._tuned[n_steps >= 4] = best  # per launch-length class (_shape_for)
        finally:
            self._tuning = tuning0
            for k, v in saved.items():
                getattr(self, k).copy_(v)
        return best

    def reset_indexed(self, idx, count):
        raise NotImplementedError("Brax families reset through a lane mask (reset(mask)) or in-kernel auto-reset")

    def reset_done(self):
        return self.reset((self.terminated | self.truncated))
