"""ctypes binding of include/carl_amd.h (libcarl_amd.so).

There is NO CPU fallback: if the HIP library is missing the import of the engine
fails loudly.  ``torch`` is imported first so that the library binds to the HIP
runtime PyTorch-ROCm already loaded (same ``libamdhip64.so.7``), which is what makes
``tensor.data_ptr()`` and ``torch.cuda.current_stream().cuda_stream`` valid on the
other side of the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

from carl_amd import build as _build

CARL_ABI_VERSION = 9
CARL_MAX_CTX_OBS = 32

# carl_family_t
CARTPOLE, PENDULUM, ACROBOT, MOUNTAINCAR, MOUNTAINCAR_CONT = range(5)
CARL_N_FAMILIES = 5
ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED = -1, -2  # CARL_ERR_* (include/carl_amd.h)
# carl_selector_t
SEL_STATIC, SEL_ROUND_ROBIN, SEL_RANDOM, SEL_HOST = range(4)
FLAG_AUTORESET = 1
FLAG_CARTPOLE_RECOMPUTE = 2
FLAG_ACROBOT_FP32 = 4
FLAG_AUTORESET_FIRST_STATE = 8
FLAG_ROLLOUT_DIRECT = 16
FLAG_BRAX_GENERIC = 32
FLAG_BRAX_FP32 = 64  # opt-in: float32 pose algebra in the substeps (include/carl_amd.h: CARL_FLAG_BRAX_FP32)
ROLLOUT_STAGED, ROLLOUT_DIRECT_SHAPE, ROLLOUT_DIRECT_FLAG = range(3)
ACTION_I32, ACTION_I64, ACTION_F32, ACTION_U8, ACTION_F16, ACTION_BF16 = range(6)

_vp = C.c_void_p


class FamilyInfo(C.Structure):
    _fields_ = [
        ("state_dim", C.c_int32), ("obs_dim", C.c_int32), ("n_features", C.c_int32),
        ("action_dim", C.c_int32), ("action_is_discrete", C.c_int32), ("n_actions", C.c_int32),
        ("max_episode_steps", C.c_int32), ("reserved", C.c_int32),
        ("action_low", C.c_float), ("action_high", C.c_float),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("family", C.c_int32), ("n_lanes", C.c_int32), ("n_contexts", C.c_int32),
        ("ctx_stride", C.c_int32), ("max_episode_steps", C.c_int32), ("selector", C.c_int32),
        ("selector_stride", C.c_int32), ("flags", C.c_int32),
        ("lane_offset", C.c_int64), ("seed", C.c_uint64),
        ("state", _vp), ("elapsed", _vp), ("ctx_idx", _vp), ("episode", _vp), ("n_calls", _vp),
        ("ep_return", _vp),
        ("ctx_table", _vp), ("ctx_obs", _vp),
        ("n_ctx_obs", C.c_int32), ("ctx_obs_feat", C.c_int32 * CARL_MAX_CTX_OBS),
        ("fin_capacity", C.c_int32),
        ("last_return", _vp), ("last_length", _vp), ("episodes_done", _vp),
        ("fin_count", _vp), ("fin_lane", _vp), ("fin_return", _vp), ("fin_length", _vp),
        ("goal_pos", _vp), ("success", _vp), ("first_state", _vp),
    ]


class StepIO(C.Structure):
    _fields_ = [
        ("action", _vp), ("action_dtype", C.c_int32), ("row_pitch", C.c_int32),
        ("obs", _vp), ("reward", _vp), ("terminated", _vp), ("truncated", _vp), ("final_obs", _vp), ("done", _vp),
        ("branch_sig", _vp),
    ]


EXPORTS = {
    "carl_abi_version": (C.c_int, []),
    "carl_last_error": (C.c_char_p, []),
    "carl_family_info": (C.c_int, [C.c_int, C.POINTER(FamilyInfo)]),
    "carl_reset": (C.c_int, [C.POINTER(Batch), _vp, _vp, _vp]),
    "carl_reset_indexed": (C.c_int, [C.POINTER(Batch), _vp, _vp, _vp, _vp]),
    "carl_step": (C.c_int, [C.POINTER(Batch), C.POINTER(StepIO), _vp]),
    "carl_rollout": (C.c_int, [C.POINTER(Batch), C.POINTER(StepIO), C.c_int32, _vp]),
    "carl_rollout_pair": (C.c_int, [C.POINTER(Batch), C.POINTER(StepIO), C.POINTER(Batch), C.POINTER(StepIO), C.c_int32, _vp]),
    "carl_rollout_variant": (C.c_int, [C.POINTER(Batch)]),
    "carl_rollout_variant_io": (C.c_int, [C.POINTER(Batch), C.POINTER(StepIO)]),
    "carl_rollout_pitch": (C.c_int32, [C.c_int32]),
    "carl_done_compact": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "carl_done_compact_scratch_elems": (C.c_int32, [C.c_int32]),
}


class CarlHipError(RuntimeError):
    pass


_lib = None


def lib_path() -> str:
    return _build.LIB_PATH


def load() -> C.CDLL:
    """Load libcarl_amd.so (building it in-tree first if hipcc is here and the
    sources are newer).  Raises if the library cannot be produced or loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    override = os.environ.get("CARL_AMD_LIB_PATH")  # kernel experiments: load another build
    if override:
        path = override
    elif os.environ.get("CARL_AMD_NO_BUILD", "0") != "1":
        try:
            path = _build.build()
        except Exception as e:  # hipcc absent / compile error
            if not os.path.exists(path):
                raise CarlHipError(
                    f"libcarl_amd.so is missing and could not be built ({e}); the engine has "
                    "no CPU fallback -- run `python -m carl_amd.build` on a machine with hipcc"
                ) from e
    if not os.path.exists(path):
        raise CarlHipError(f"{path} not found; run `python -m carl_amd.build`")
    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.carl_abi_version()
    if v != CARL_ABI_VERSION:
        raise CarlHipError(f"libcarl_amd ABI version {v} != binding {CARL_ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        msg = load().carl_last_error()
        raise CarlHipError(f"libcarl_amd call failed ({code}): {msg.decode() if msg else ''}")


def family_info(family: int) -> FamilyInfo:
    info = FamilyInfo()
    check(load().carl_family_info(family, C.byref(info)))
    return info


# ---- Brax-locomotion families (carl_brax_sys_t, include/carl_amd.h) ----------------------
BRAX_MAX_LINKS, BRAX_MAX_DOF, BRAX_MAX_Q, BRAX_MAX_ACT, BRAX_MAX_COLL, BRAX_MAX_CTX_MASS = 16, 24, 32, 24, 32, 16
BRAX_MAX_PAIR = 8
(BRAX_ANT, BRAX_HALFCHEETAH, BRAX_HUMANOID, BRAX_HOPPER, BRAX_WALKER2D, BRAX_INVERTED_PENDULUM, BRAX_HUMANOIDSTANDUP, BRAX_INVERTED_DOUBLE_PENDULUM,
 BRAX_REACHER, BRAX_PUSHER) = range(10)
BRAX_LINK_STATE = 13
BRAX_LINK_RECORD = 20  # floats per (env, link) in HBM: pose head 7 | pose tail 7 | velocities 6 (include/carl_amd.h, ABI 8)
_f, _i = C.c_float, C.c_int32


class BraxCtxMap(C.Structure):
    _fields_ = [
        ("gravity", _i), ("friction", _i), ("elasticity", _i), ("ang_damping", _i),
        ("joint_stiffness_scale", _i), ("target_distance", _i), ("target_direction", _i), ("target_radius", _i),
        ("n_mass", _i),
        ("mass_row", _i * BRAX_MAX_CTX_MASS), ("mass_link", _i * BRAX_MAX_CTX_MASS),
        ("mass_nominal", _f * BRAX_MAX_CTX_MASS), ("mass_ratio_floor", _f * BRAX_MAX_CTX_MASS),
        ("mass_ratio_floor_multi", _f * BRAX_MAX_CTX_MASS),
        ("goal_position", _i * 3),
    ]


class BraxSys(C.Structure):
    _fields_ = [
        ("env_kind", _i), ("n_links", _i), ("n_q", _i), ("n_dof", _i), ("n_act", _i), ("n_coll", _i),
        ("n_frames", _i), ("obs_dim", _i), ("max_episode_steps", _i), ("terminate_when_unhealthy", _i),
        ("exclude_current_positions", _i), ("reserved", _i),
        ("dt", _f), ("gravity_z", _f), ("vel_damping", _f), ("ang_damping", _f), ("baumgarte_erp", _f),
        ("elasticity", _f), ("friction", _f),
        ("healthy_z_lo", _f), ("healthy_z_hi", _f), ("healthy_reward", _f), ("ctrl_cost_weight", _f),
        ("forward_reward_weight", _f), ("reset_noise_scale", _f), ("reset_vel_scale", _f),
        ("parent", _i * BRAX_MAX_LINKS), ("n_link_dof", _i * BRAX_MAX_LINKS),
        ("q_start", _i * BRAX_MAX_LINKS), ("dof_start", _i * BRAX_MAX_LINKS),
        ("link_pos", (_f * 3) * BRAX_MAX_LINKS), ("link_rot", (_f * 4) * BRAX_MAX_LINKS),
        ("joint_pos", (_f * 3) * BRAX_MAX_LINKS), ("joint_rot", (_f * 4) * BRAX_MAX_LINKS),
        ("com", (_f * 3) * BRAX_MAX_LINKS), ("mass", _f * BRAX_MAX_LINKS),
        ("inv_inertia", (_f * 3) * BRAX_MAX_LINKS),
        ("k_pos", _f * BRAX_MAX_LINKS), ("k_vel", _f * BRAX_MAX_LINKS),
        ("k_limit", _f * BRAX_MAX_LINKS), ("k_ang_damp", _f * BRAX_MAX_LINKS),
        ("dof_lo", _f * BRAX_MAX_DOF), ("dof_hi", _f * BRAX_MAX_DOF),
        ("dof_damping", _f * BRAX_MAX_DOF), ("dof_stiffness", _f * BRAX_MAX_DOF),
        ("act_dof", _i * BRAX_MAX_ACT), ("act_gear", _f * BRAX_MAX_ACT),
        ("act_lo", _f * BRAX_MAX_ACT), ("act_hi", _f * BRAX_MAX_ACT),
        ("coll_link", _i * BRAX_MAX_COLL), ("coll_pos", (_f * 3) * BRAX_MAX_COLL),
        ("coll_radius", _f * BRAX_MAX_COLL),
        ("init_q", _f * BRAX_MAX_Q),
        ("goal_mode", _i), ("goal_obs_idx", _i * 2), ("goal_dt", _f),
        ("n_slide", _i * BRAX_MAX_LINKS), ("dof_sign3", _f * BRAX_MAX_LINKS),
        ("reset_vel_uniform", _i), ("reward_on_com", _i), ("obs_extended", _i), ("healthy_q_index", _i),
        ("healthy_q_lo", _f), ("healthy_q_hi", _f), ("obs_qd_clip", _f), ("lanes_per_env", _i),
        ("reward_height", _i), ("obs_trig_from", _i),
        ("tip_link", _i), ("tip_offset", _f * 3), ("tip_x_weight", _f), ("tip_height", _f), ("tip_min_height", _f),
        ("tip_vel_weight", _f * 2), ("tip_vel_dof", _i * 2),
        ("target_link", _i), ("target_max_dist", _f),
        ("push_link", _i), ("push_goal", _f * 3), ("push_near_weight", _f), ("push_min_dist", _f),
        ("push_lo", _f * 2), ("push_hi", _f * 2),
        ("n_pair", _i), ("pair_link", _i), ("pair_pos", (_f * 3) * BRAX_MAX_PAIR), ("pair_radius", _f * BRAX_MAX_PAIR),
        ("pair_obj_radius", _f), ("pair_obj_half", _f), ("pair_k", _f), ("pair_c", _f),
        ("pair_ct", _f), ("plane_z", _f), ("obj_support", _i), ("reserved2", _i),
        ("slide_axis", ((_f * 3) * 2) * BRAX_MAX_LINKS),
        ("ctx", BraxCtxMap),
    ]


EXPORTS.update({
    "carl_brax_reset": (C.c_int, [C.POINTER(Batch), _vp, C.POINTER(BraxSys), _vp, _vp, _vp]),
    "carl_brax_step": (C.c_int, [C.POINTER(Batch), _vp, C.POINTER(BraxSys), C.POINTER(StepIO), _vp]),
    "carl_brax_rollout": (C.c_int, [C.POINTER(Batch), _vp, C.POINTER(BraxSys), C.POINTER(StepIO), C.c_int32, _vp]),
    "carl_brax_lane_widths": (C.c_int, [C.POINTER(BraxSys), C.c_uint32, _vp, C.c_int32]),
    "carl_brax_model_is_planar": (C.c_int, [C.POINTER(BraxSys)]),
    "carl_brax_fragment_plan": (C.c_int, [C.c_int32] * 6 + [_vp, C.c_int32]),
})


# ---- context sets on the device (include/carl_amd.h: carl_feature_spec_t) --------------------
MAX_CHOICES = 32
FEAT_CONSTANT, FEAT_UNIFORM_FLOAT, FEAT_NORMAL_FLOAT, FEAT_UNIFORM_INT, FEAT_CATEGORICAL = range(5)


class FeatureSpec(C.Structure):
    _fields_ = [
        ("kind", _i), ("n_choices", _i), ("log_scale", _i), ("reserved", _i),
        ("lower", _f), ("upper", _f), ("mu", _f), ("sigma", _f), ("value", _f), ("reserved_f", _f),
        ("choices", _f * MAX_CHOICES),
    ]


EXPORTS.update({
    "carl_sample_contexts": (C.c_int, [_vp, C.POINTER(FeatureSpec), C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                       C.c_uint64, _vp, _vp]),
    "carl_verify_contexts": (C.c_int, [_vp, C.POINTER(FeatureSpec), C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
})
