"""C-ABI library: loads, exports every symbol include/carl_amd.h declares, and the ctypes
structs have the C layout.  CPU-only: nothing here launches a kernel."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "carl_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(carl_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from carl_amd import _lib

    lib = _lib.load()
    names = declared_functions()
    assert "carl_step" in names and "carl_rollout" in names and len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/carl_amd.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes binding and header disagree"
    assert lib.carl_abi_version() == _lib.CARL_ABI_VERSION


def test_action_dtype_codes_match_the_header():
    """CARL_ACTION_* of include/carl_amd.h == the binding's constants (ABI 7 added U8, a rollout-only input format)"""
    from carl_amd import _lib

    src = open(HEADER).read()
    m = re.search(r"enum \{ (CARL_ACTION_I32[^}]*)\};", re.sub(r"/\*.*?\*/", "", src, flags=re.S))
    codes = {k.strip(): int(v) for k, v in (item.split("=") for item in m.group(1).split(","))}
    assert codes == {"CARL_ACTION_I32": _lib.ACTION_I32, "CARL_ACTION_I64": _lib.ACTION_I64,
                     "CARL_ACTION_F32": _lib.ACTION_F32, "CARL_ACTION_U8": _lib.ACTION_U8,
                     "CARL_ACTION_F16": _lib.ACTION_F16, "CARL_ACTION_BF16": _lib.ACTION_BF16}
    assert int(re.search(r"#define CARL_ABI_VERSION (\d+)", src).group(1)) == _lib.CARL_ABI_VERSION == 9


def test_family_info_is_host_side():
    from carl_amd import _lib

    want = {0: (4, 4, 8, 1, 2, 500), 1: (2, 3, 7, 0, 0, 200), 2: (4, 6, 14, 1, 3, 500), 3: (2, 2, 11, 1, 3, 200),
            4: (2, 2, 10, 0, 0, 999)}
    for fam, w in want.items():
        i = _lib.family_info(fam)
        assert (i.state_dim, i.obs_dim, i.n_features, i.action_is_discrete, i.n_actions, i.max_episode_steps) == w
    with pytest.raises(_lib.CarlHipError):
        _lib.family_info(99)
    assert b"unknown family" in _lib.load().carl_last_error()


def test_invalid_arguments_are_reported_not_crashed():
    from carl_amd import _lib

    lib = _lib.load()
    assert lib.carl_reset(None, None, None, None) == -1
    b = _lib.Batch()
    b.family = 1
    b.n_lanes = 4
    assert lib.carl_step(C.byref(b), None, None) == -1  # n_contexts == 0
    assert b"n_contexts" in lib.carl_last_error()
    assert lib.carl_done_compact(None, None, -1, None, None, None, None) == -1
    assert lib.carl_done_compact_scratch_elems(0) == 1 and lib.carl_done_compact_scratch_elems(1025) == 2


def test_ctypes_struct_layout_matches_c(tmp_path):
    """compile a tiny C program against the header and compare sizeof/offsetof"""
    from carl_amd import _lib

    prog = tmp_path / "layout.c"
    fields_b = [f[0] for f in _lib.Batch._fields_]
    fields_io = [f[0] for f in _lib.StepIO._fields_]
    fields_fi = [f[0] for f in _lib.FamilyInfo._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){",
             'printf("%zu %zu %zu\\n", sizeof(carl_batch_t), sizeof(carl_step_io_t), sizeof(carl_family_info_t));']
    for f in fields_b:
        lines.append(f'printf("%zu\\n", offsetof(carl_batch_t, {f}));')
    for f in fields_io:
        lines.append(f'printf("%zu\\n", offsetof(carl_step_io_t, {f}));')
    for f in fields_fi:
        lines.append(f'printf("%zu\\n", offsetof(carl_family_info_t, {f}));')
    lines.append("return 0;}")
    prog.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(prog)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes, offs = list(map(int, out[:3])), list(map(int, out[3:]))
    assert sizes == [C.sizeof(_lib.Batch), C.sizeof(_lib.StepIO), C.sizeof(_lib.FamilyInfo)]
    want = ([getattr(_lib.Batch, f).offset for f in fields_b] + [getattr(_lib.StepIO, f).offset for f in fields_io]
            + [getattr(_lib.FamilyInfo, f).offset for f in fields_fi])
    assert offs == want


def test_engine_refuses_cpu():
    """no CPU fallback: asking for a CPU device is an error, not a silent slow path"""
    from carl_amd import _lib
    from carl_amd.engine import VecEngine

    with pytest.raises(_lib.CarlHipError):
        VecEngine(1, [[8.0, 0.05, 10, 1, 1, 3.14, 1]], 4, device="cpu")


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: no module under carl_amd/ may reference it"""
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "carl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hip.h", ".h")):
                txt = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt and f.endswith(".py"):
                    bad.append(f)
    assert not bad, bad
    code = "import sys; import carl_amd.envs, carl_amd.engine, carl_amd.distributed; " \
           "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_missing_library_fails_loudly():
    """the product path has no fallback: without the HIP library every entry point is an error"""
    code = ("import os; os.environ['CARL_AMD_LIB_PATH'] = '/nonexistent/libcarl_amd.so'\n"
            "from carl_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.CarlHipError as e:\n    print('LOUD', e)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, check=True).stdout
    assert "LOUD" in out and "not found" in out
    code = ("import os; os.environ['CARL_AMD_LIB_PATH'] = '/nonexistent/libcarl_amd.so'\n"
            "from carl_amd.envs import CARLPendulum\n"
            "try:\n    CARLPendulum()\nexcept Exception as e:\n    print(type(e).__name__)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, check=True).stdout
    assert "CarlHipError" in out


def test_brax_and_sampler_entry_points_validate_arguments():
    from carl_amd import _lib
    from carl_amd.envs.brax.models import ant_sys

    lib = _lib.load()
    assert lib.carl_brax_reset(None, None, None, None, None, None) == -1
    s = ant_sys()
    b = _lib.Batch()
    b.n_lanes, b.n_contexts, b.ctx_stride = 4, 1, 1
    assert lib.carl_brax_step(C.byref(b), None, C.byref(s), None, None) == -1  # sys_dev NULL
    widths = (C.c_int32 * 16)()
    n = lib.carl_brax_lane_widths(C.byref(s), 0, widths, 16)
    assert [widths[i] for i in range(n)] == [9, 16]  # one lane per link (Ant: 9 links) or wider
    assert lib.carl_brax_lane_widths(None, 0, widths, 16) == 0
    # ABI 9: the widths are those of the kernels a STEP launch of the batch takes -- a planar model under
    # CARL_FLAG_BRAX_GENERIC runs the multi-hinge kernels (its autotune probed 7 / 8, all one kernel)
    from carl_amd.envs.brax.models import halfcheetah_sys

    hc = halfcheetah_sys()
    n = lib.carl_brax_lane_widths(C.byref(hc), 0, widths, 16)
    assert [widths[i] for i in range(n)] == [7, 8, 9, 16]
    n = lib.carl_brax_lane_widths(C.byref(hc), _lib.FLAG_BRAX_GENERIC, widths, 16)
    assert [widths[i] for i in range(n)] == [11, 16]
    # ABI 9: row pitch of a rollout's buffers
    assert [lib.carl_rollout_pitch(k) for k in (-3, 0, 1, 10, 16, 17, 65537)] == [0, 0, 16, 16, 16, 32, 65552]
    cb = _lib.Batch()
    cb.family, cb.n_lanes = _lib.CARTPOLE, 10
    io = _lib.StepIO()
    assert lib.carl_rollout_variant(C.byref(cb)) == _lib.ROLLOUT_DIRECT_SHAPE
    assert lib.carl_rollout_variant_io(C.byref(cb), C.byref(io)) == _lib.ROLLOUT_DIRECT_SHAPE  # dense rows
    io.row_pitch = 16
    assert lib.carl_rollout_variant_io(C.byref(cb), C.byref(io)) == _lib.ROLLOUT_STAGED
    cb.n_lanes, io.row_pitch = 65537, 65552
    assert lib.carl_rollout_variant_io(C.byref(cb), C.byref(io)) == _lib.ROLLOUT_STAGED
    io.row_pitch = 65536
    assert lib.carl_rollout_variant_io(C.byref(cb), C.byref(io)) == _lib.ERR_INVALID_ARGUMENT  # pitch < n_lanes
    # a pitch beyond the padded one is a view into a wider array: the columns next to the lanes belong to someone else, so
    # the staged kernel (which writes whole 16-byte pieces) only runs when there is no padding to write
    cb.n_lanes, io.row_pitch = 1000, 2080
    assert lib.carl_rollout_variant_io(C.byref(cb), C.byref(io)) == _lib.ROLLOUT_DIRECT_SHAPE
    cb.n_lanes, io.row_pitch = 1024, 2048
    assert lib.carl_rollout_variant_io(C.byref(cb), C.byref(io)) == _lib.ROLLOUT_STAGED
    cb.n_lanes = 65537
    cb.flags = _lib.FLAG_ROLLOUT_DIRECT
    io.row_pitch = 65552
    assert lib.carl_rollout_variant_io(C.byref(cb), C.byref(io)) == _lib.ROLLOUT_DIRECT_FLAG
    spec = (_lib.FeatureSpec * 1)()
    spec[0].kind = 99
    assert lib.carl_sample_contexts(C.addressof(spec), spec, 1, 4, 4, 0, 0, 1, None) == -1  # table "pointer" 1, kind 99
    assert b"unknown kind" in lib.carl_last_error()
    spec[0].kind, spec[0].lower, spec[0].upper = _lib.FEAT_UNIFORM_FLOAT, 2.0, 1.0
    assert lib.carl_sample_contexts(C.addressof(spec), spec, 1, 4, 4, 0, 0, 1, None) == -1
    assert b"lower" in lib.carl_last_error()


def test_first_state_flag_needs_its_buffer():
    """CARL_FLAG_AUTORESET_FIRST_STATE without carl_batch_t::first_state is refused (no silent fall-back to the
    re-draw rule); checked on the host before anything is launched, so it runs without a GPU"""
    import ctypes as C

    from carl_amd import _lib
    from carl_amd.envs.brax.models import ant_sys

    lib = _lib.load()
    b = _lib.Batch()
    b.n_lanes, b.n_contexts, b.ctx_stride = 8, 1, 1
    for f in ("state", "elapsed", "ctx_idx", "episode", "n_calls", "ep_return", "ctx_table"):
        setattr(b, f, 0x1000)  # never dereferenced: validation fails first
    b.flags = _lib.FLAG_AUTORESET | _lib.FLAG_AUTORESET_FIRST_STATE
    s = ant_sys(["gravity", "friction", "elasticity", "ang_damping", "mass_torso", "viscosity"])
    rc = lib.carl_brax_reset(C.byref(b), 0x1000, C.byref(s), None, None, None)
    assert rc == -1 and b"first_state" in lib.carl_last_error()


def test_rollout_variant_is_answered_on_the_host():
    """carl_rollout_variant (ABI 5): which kernel a fused rollout of this batch launches -- pure host logic"""
    from carl_amd import _lib

    lib = _lib.load()
    b = _lib.Batch()
    b.n_lanes = 65536
    assert lib.carl_rollout_variant(C.byref(b)) == _lib.ROLLOUT_STAGED
    b.n_lanes = 65520  # a multiple of 16: still the staged kernel (ragged last workgroup)
    assert lib.carl_rollout_variant(C.byref(b)) == _lib.ROLLOUT_STAGED
    b.n_lanes = 65000
    assert lib.carl_rollout_variant(C.byref(b)) == _lib.ROLLOUT_DIRECT_SHAPE
    b.n_lanes = 65536
    b.flags = _lib.FLAG_ROLLOUT_DIRECT
    assert lib.carl_rollout_variant(C.byref(b)) == _lib.ROLLOUT_DIRECT_FLAG
    assert lib.carl_rollout_variant(None) == -1 and b"NULL" in lib.carl_last_error()
    b.family = -1  # a Brax batch: one rollout kernel, the question does not apply
    assert lib.carl_rollout_variant(C.byref(b)) == -1 and b"classic-control" in lib.carl_last_error()


def test_planar_model_query_is_answered_on_the_host():
    """carl_brax_model_is_planar: which models step / rollout launches advance with the planar substep"""
    from carl_amd import _lib
    from carl_amd.envs import brax as BX
    from carl_amd.envs.brax import models

    lib = _lib.load()
    want = {"halfcheetah": 1, "hopper": 1, "walker2d": 1, "ant": 0, "humanoid": 0, "humanoidstandup": 0,
            "inverted_pendulum": 0, "inverted_double_pendulum": 0, "reacher": 0, "pusher": 0}
    classes = {c.env_name: c for c in (getattr(BX, n) for n in dir(BX)) if isinstance(c, type) and hasattr(c, "env_name")}
    for name, flag in want.items():
        names = list(classes[name].get_context_features().keys())
        s = models.SYSTEMS[name](names)
        assert lib.carl_brax_model_is_planar(C.byref(s)) == flag, name
    assert lib.carl_brax_model_is_planar(None) == 0 and b"NULL" in lib.carl_last_error()


def _plan(lib, n_groups, n_wg, n_waves, T, wg, wave):
    out = (C.c_int32 * (5 * 1024))()
    k = lib.carl_brax_fragment_plan(n_groups, n_wg, n_waves, T, wg, wave, out, 1024)
    assert 0 <= k <= 1024
    return [tuple(out[5 * i: 5 * i + 5]) for i in range(k)]


def test_brax_fragment_schedule_covers_every_group_step_once_and_cannot_deadlock():
    """The (group, step-range) schedule of a Brax step / rollout launch (include/carl_amd.h: carl_brax_fragment_plan --
    the integer functions the kernel itself runs).  For many (groups, workgroups, wavefronts, steps): every group-step
    is run exactly once; a group is split between at most two wavefronts of ONE workgroup, head before tail; a
    wavefront raises its hand-over flag in its FIRST fragment and waits only in its LAST, for the PREVIOUS wavefront
    (so no wait can depend on a later one: no deadlock among co-resident wavefronts); pieces of a workgroup with more
    groups than wavefronts differ by at most one step (balance)."""
    from carl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(0)
    cases = [(4682, 256, 12, 20), (6554, 256, 8, 20), (4286, 256, 12, 7), (7500, 256, 12, 7), (3, 1, 4, 5), (13, 1, 12, 2),
             (100, 7, 3, 1), (18, 1, 12, 4), (12, 1, 12, 9), (5, 5, 1, 3)]
    cases += [(int(rng.integers(1, 400)), int(rng.integers(1, 9)), int(rng.integers(1, 13)), int(rng.integers(1, 40)))
              for _ in range(200)]
    for n_groups, n_wg, n_waves, T in cases:
        n_wg = min(n_wg, n_groups)
        seen = np.zeros((n_groups, T), np.int32)
        for wg in range(n_wg):
            g_lo, g_hi = wg * n_groups // n_wg, (wg + 1) * n_groups // n_wg
            G = g_hi - g_lo
            plans = [_plan(lib, n_groups, n_wg, n_waves, T, wg, w) for w in range(n_waves)]
            lengths = []
            for w, frags in enumerate(plans):
                lengths.append(sum(t1 - t0 for _, t0, t1, _, _ in frags))
                for i, (g, t0, t1, wait, signal) in enumerate(frags):
                    assert g_lo <= g < g_hi and 0 <= t0 < t1 <= T, (n_groups, n_wg, n_waves, T, wg, w, frags)
                    seen[g, t0:t1] += 1
                    if signal:  # a head: first thing the wavefront does, starts at step 0, continues in wavefront w + 1
                        assert i == 0 and t0 == 0 and t1 < T and w + 1 < n_waves
                        nxt = plans[w + 1][-1]
                        assert nxt[0] == g and nxt[1] == t1 and nxt[2] == T and nxt[3] == 1
                    if wait:    # a tail: last thing the wavefront does, ends at T, its head ran in wavefront w - 1
                        assert i == len(frags) - 1 and t1 == T and t0 > 0 and w > 0
                        prv = plans[w - 1][0]
                        assert prv[0] == g and prv[1] == 0 and prv[2] == t0 and prv[4] == 1
                    if not wait and not signal:
                        assert (t0, t1) == (0, T)
                # timing: the head this wavefront waits for (t0 steps long, run first by w - 1) is over before this
                # wavefront gets to its last fragment, if all wavefronts advance at the same rate
                if frags and frags[-1][3]:
                    assert frags[-1][1] <= lengths[-1] - (T - frags[-1][1])
            if G > n_waves:
                assert max(lengths) - min(lengths) <= 1 and sum(lengths) == G * T
            else:
                assert sorted(lengths, reverse=True) == [T] * G + [0] * (n_waves - G)
        assert (seen == 1).all(), (n_groups, n_wg, n_waves, T)
    assert lib.carl_brax_fragment_plan(0, 1, 1, 1, 0, 0, None, 0) == -1
    assert lib.carl_brax_fragment_plan(10, 2, 4, 5, 2, 0, None, 0) == -1
