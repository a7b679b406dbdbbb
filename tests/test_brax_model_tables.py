"""The Brax model tables, decoupled.

`carl_amd/envs/brax/models.py` (hand-written Python filling ``carl_brax_sys_t``) feeds both the HIP kernels and the
CPU restatement, so no parity test can see a wrong number in it.  Here the same facts come a second way -- MJCF text
under oracle/mjcf/ (the assets the reference names, written out from upstream memory) read by oracle/mjcf_tables.py,
which shares no code with models.py -- and are

(a) PINNED AGAINST THE REFERENCE: the link masses the reader derives from the geometry (capsule / sphere volumes x
    density, `settotalmass`) equal the numbers the reference holds as context defaults
    (carl/envs/brax/carl_halfcheetah.py:37-57, carl_humanoid.py:37-75; read by tests/golden/make_feature_table_golden.py
    into tests/golden/context_feature_tables.json) and the legacy torso masses of
    docs/source/environments/data/context_definitions/CARL{Halfcheetah,Humanoid}.csv -- to 1e-7: every radius and
    length of those 18 links is thereby tied to a number from the reference tree;
(b) compared with models.py field by field: link tree, frames, joint axes / anchors / ranges / stiffness / damping,
    actuators (joint, gear, control range, ORDER), colliders (which spheres on which link), centres of mass, q layout
    and reset pose.
"""
import json
import math
import os

import numpy as np
import pytest

from carl_amd.envs import brax as BX
from carl_amd.envs.brax import models
from oracle import mjcf_tables as M

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "context_feature_tables.json")
CLASS = {"ant": "CARLBraxAnt", "halfcheetah": "CARLBraxHalfcheetah", "humanoid": "CARLBraxHumanoid",
         "humanoidstandup": "CARLBraxHumanoidStandup"}
ASSET = {"humanoidstandup": "humanoid"}  # humanoidstandup.xml is the Humanoid body lying on its back: same links, other qpos0


def _table(name):
    names = list(getattr(BX, CLASS[name]).get_context_features().keys())
    return models.SYSTEMS[name](names), names


def _rot(q, v):
    return M._qrot(np.asarray(q, dtype=np.float64), v)


def test_masses_from_the_geometry_equal_the_numbers_the_reference_holds():
    ref = json.load(open(GOLDEN))
    # legacy torso masses (density 1000, BEFORE settotalmass): docs/.../CARLHalfcheetah.csv, CARLHumanoid.csv
    legacy_torso = {"halfcheetah": 9.457333, "humanoid": 8.907463}
    n_pinned = 0
    for name in ("halfcheetah", "humanoid"):
        m = M.load(name)
        want = {f["name"][5:]: float(f["default_value"]) for f in ref[CLASS[name]] if f["name"].startswith("mass_")}
        assert set(want) == {l.name for l in m.links}  # the reference's link names ARE the asset's body names
        for l in m.links:
            if l.name == "torso":  # CARL's default is 10 for every env (carl_halfcheetah.py:37), not the asset's value
                raw = sum(g.volume * g.density for g in l.geoms)
                assert abs(raw - legacy_torso[name]) < 2e-6, (name, raw)
            else:
                assert abs(l.mass - want[l.name]) < 2e-7 * want[l.name] + 1e-7, (name, l.name, l.mass, want[l.name])
            n_pinned += 1
    assert n_pinned == 18
    # Halfcheetah: `settotalmass` = 14 (the six link masses only come out right with it)
    assert abs(sum(l.mass for l in M.load("halfcheetah").links) - 14.0) < 1e-9


def _dominated(spheres):
    """drop spheres wholly inside another sphere of the same link (they can never touch the plane first)"""
    keep = []
    for i, (c, r) in enumerate(spheres):
        if not any(j != i and np.linalg.norm(np.asarray(c) - np.asarray(c2)) + r <= r2 + 1e-12 for j, (c2, r2) in enumerate(spheres)):
            keep.append((c, r))
    return keep


@pytest.mark.parametrize("name", ["ant", "halfcheetah", "humanoid", "humanoidstandup"])
def test_model_table_equals_the_independent_restatement_field_by_field(name):
    # NOTE: oracle/mjcf/*.xml were written from memory by the author of models.py.  For the GEOMETRY this
    # comparison is anchored outside the build (the link masses the capsule volumes give equal the reference's context
    # defaults: the test above); for the non-geometric fields -- joint stiffness / damping / ranges, gears -- it only says
    # that two from-memory restatements agree (round 4 changed Humanoid's knee stiffness and left-hip range on that
    # basis).  When brax's assets become available: diff oracle/mjcf/*.xml against them and record a hash.
    s, _ = _table(name)
    m = M.load(ASSET.get(name, name))
    L = len(m.links)
    assert (s.n_links, s.n_q, s.n_dof, s.n_act) == (L, m.n_q, m.n_dof, len(m.actuators))
    qi = di = 0
    for i, l in enumerate(m.links):
        assert s.parent[i] == l.parent, (name, l.name)
        np.testing.assert_allclose(list(s.link_pos[i]), l.pos if l.parent >= 0 or name != "ant" else list(s.link_pos[i]), atol=1e-7)
        np.testing.assert_allclose(list(s.link_rot[i]), l.quat, atol=1e-7)
        np.testing.assert_allclose(list(s.com[i]), l.com, atol=2e-6, err_msg=f"{name}/{l.name} com")
        assert (s.q_start[i], s.dof_start[i]) == (qi, di), (name, l.name)
        kinds = [j.kind for j in l.joints]
        if kinds == ["free"]:
            assert s.n_link_dof[i] == 6 and s.parent[i] == -1
            qi, di = qi + 7, di + 6
            continue
        slides = [j for j in l.joints if j.kind == "slide"]
        hinges = [j for j in l.joints if j.kind == "hinge"]
        assert kinds == ["slide"] * len(slides) + ["hinge"] * len(hinges)  # q order: slides, then the hinges
        assert (s.n_slide[i], s.n_link_dof[i]) == (len(slides), len(l.joints)), (name, l.name)
        for j in l.joints:  # one anchor per link in these assets
            np.testing.assert_allclose(list(s.joint_pos[i]), j.pos, atol=1e-7, err_msg=f"{name}/{j.name} anchor")
        for k, j in enumerate(slides):
            np.testing.assert_allclose(list(s.slide_axis[i][k]), j.axis, atol=1e-7)
        # hinge axes = the joint frame's x, y, sign3 * z (body frame)
        basis = [(1, 0, 0), (0, 1, 0), (0, 0, float(s.dof_sign3[i]))]
        for k, j in enumerate(hinges):
            np.testing.assert_allclose(_rot(list(s.joint_rot[i]), basis[k]), j.axis, atol=1e-6, err_msg=f"{name}/{j.name} axis")
        for k, j in enumerate(l.joints):
            d = di + k
            if math.isinf(j.lo):
                assert s.dof_lo[d] <= -1e8 and s.dof_hi[d] >= 1e8, (name, j.name)
            else:
                assert abs(s.dof_lo[d] - j.lo) < 1e-6 and abs(s.dof_hi[d] - j.hi) < 1e-6, (name, j.name, s.dof_lo[d], j.lo, s.dof_hi[d], j.hi)
            assert abs(s.dof_stiffness[d] - j.stiffness) < 1e-6, (name, j.name, "stiffness", s.dof_stiffness[d], j.stiffness)
            assert abs(s.dof_damping[d] - j.damping) < 1e-6, (name, j.name, "damping", s.dof_damping[d], j.damping)
        qi, di = qi + len(l.joints), di + len(l.joints)
    # actuators: same joints in the same ORDER (the action vector's layout), control ranges, and the gears of the System
    # the brax env class steps under backend="spring" (carl_brax_env.py:117): the MJCF's, with the env constructor's
    # spring-backend override where it has one (oracle/mjcf_tables.py: SPRING_BACKEND; round 6) -- and that branch's
    # timestep / n_frames
    env = M.spring_env(name)
    for k, (jn, gear, lo, hi) in enumerate(env["actuators"]):
        assert s.act_dof[k] == m.dof_of(jn), (name, k, jn)
        assert (s.act_gear[k], s.act_lo[k], s.act_hi[k]) == pytest.approx((gear, lo, hi), abs=1e-6), (name, jn)
    assert abs(s.dt - env["dt"]) < 1e-9 and s.n_frames == env["n_frames"], (name, s.dt, s.n_frames)
    # colliders: per link the same set of spheres (capsule = its two end spheres)
    got = {i: [] for i in range(L)}
    for k in range(s.n_coll):
        got[s.coll_link[k]].append((tuple(round(float(x), 6) for x in s.coll_pos[k]), round(float(s.coll_radius[k]), 6)))
    for i, l in enumerate(m.links):
        want = sorted((tuple(round(float(x), 6) for x in c), round(r, 6)) for c, r in _dominated(l.spheres))
        assert sorted(_dominated(got[i])) == want, (name, l.name, sorted(got[i]), want)
    # reset pose
    q0 = np.array([s.init_q[i] for i in range(s.n_q)])
    if name in ASSET:  # same body, its own reset pose (lying on its back, low): only the joint coordinates are compared
        np.testing.assert_allclose(q0[7:], m.init_q[7:], atol=1e-7)
        assert q0[2] < 0.5 and abs(np.linalg.norm(q0[3:7]) - 1.0) < 1e-6
    elif m.links[0].joints[0].kind == "free":
        np.testing.assert_allclose(q0[2:], m.init_q[2:], atol=1e-7)  # (x, y start at 0)
    else:
        np.testing.assert_allclose(q0, 0.0)
    assert abs(s.goal_dt - m.timestep) < 1e-9  # the goal wrapper integrates with the RAW MJCF timestep (Quirk B3)


def test_which_colliders_each_model_carries():
    """spheres vs the plane z = 0 only; a capsule collides as its two end spheres (documented in DESIGN.md 5.1)"""
    want = {"ant": 21, "halfcheetah": 16, "humanoid": 29}
    for name, n in want.items():
        s, _ = _table(name)
        assert s.n_coll == n, (name, s.n_coll)
        m = M.load(name)
        assert sum(len(_dominated(l.spheres)) for l in m.links) == n, name


@pytest.mark.parametrize("name,cls", [("hopper", "CARLBraxHopper"), ("walker2d", "CARLBraxWalker2d")])
def test_planar_models_capsules_reproduce_the_reference_masses(name, cls):
    """Hopper / Walker2d have no MJCF restatement here, but their tables carry every capsule as its
    two end spheres (consecutive colliders of one link, same radius): density 1000 x (pi r^2 L + 4/3 pi r^3) per link
    must equal the context default the reference holds for that link (carl/envs/brax/carl_hopper.py,
    carl_walker2d.py via tests/golden/context_feature_tables.json) -- which pins the radii and lengths of the table."""
    ref = json.load(open(GOLDEN))
    names = list(getattr(BX, cls).get_context_features().keys())
    s = models.SYSTEMS[name](names)
    want = {f["name"][5:]: float(f["default_value"]) for f in ref[cls] if f["name"].startswith("mass_")}
    cm = s.ctx
    link_of = {names[cm.mass_row[k]][5:]: cm.mass_link[k] for k in range(cm.n_mass)}
    assert set(link_of) == set(want)
    vol = {i: 0.0 for i in range(s.n_links)}
    assert s.n_coll % 2 == 0
    for k in range(0, s.n_coll, 2):
        assert s.coll_link[k] == s.coll_link[k + 1] and s.coll_radius[k] == s.coll_radius[k + 1]
        r = float(s.coll_radius[k])
        length = float(np.linalg.norm(np.array(list(s.coll_pos[k])) - np.array(list(s.coll_pos[k + 1]))))
        vol[s.coll_link[k]] += math.pi * r * r * length + 4.0 / 3.0 * math.pi * r ** 3
    n = 0
    for link, i in link_of.items():
        if link == "torso":  # CARL's default is 10, not the asset's (3.6651914 for both: a 0.4 m capsule of radius 0.05)
            assert abs(1000.0 * vol[i] - 3.6651914) < 1e-5
            continue
        assert abs(1000.0 * vol[i] - want[link]) < 2e-6 * want[link], (name, link, 1000.0 * vol[i], want[link])
        n += 1
    assert n == (3 if name == "hopper" else 6)


def test_viscosity_rule_of_the_reference_is_available():
    """Quirk B2: the reference writes `viscosity` into `ang_damping` after the `ang_damping` line
    (carl/envs/brax/carl_brax_env.py:276-279).  Default here: observed only; "reference": the literal rule."""
    s, names = _table("ant")
    row_ad, row_vi = names.index("ang_damping"), names.index("viscosity")
    assert s.ctx.ang_damping == row_ad
    models.apply_viscosity_rule(s, names, "observed")
    assert s.ctx.ang_damping == row_ad
    models.apply_viscosity_rule(s, names, "reference")
    assert s.ctx.ang_damping == row_vi
    compat = models.SYSTEMS["ant"](names, reference_compat=True)  # physics contexts ignored: nothing to overwrite
    models.apply_viscosity_rule(compat, names, "reference")
    assert compat.ctx.ang_damping == -1
    with pytest.raises(ValueError):
        models.apply_viscosity_rule(s, names, "fluid")
