"""Context-feature tables of the ten Brax families, as data.

One line per family: the columns of its context table, in order (the order is part of the contract:
``ContextTable`` columns, ``carl_brax_ctx_map_t`` rows and the ``context`` observation follow it).
``mass_<link>=<default>`` declares a link-mass feature; the other tokens name declarations shared by
every family (``SHARED``); ``GOAL`` / ``GOAL_POSITION`` expand to the goal features.  The values are
the reference's (carl/envs/brax/carl_<family>.py ``get_context_features``), pinned against it by
``tests/test_feature_tables.py``; ``joint_stiffness`` is this build's extension (SURVEY.md Quirk B4): it is NOT in
the default tables, only in those of the opt-in ``...Stiffness`` classes, after the reference's columns."""
from __future__ import annotations

import math

from carl_amd.context.context_space import CategoricalContextFeature, ContextFeature, UniformFloatContextFeature

INF = math.inf
# compass codes of the goal wrapper (carl/envs/brax/brax_walker_goal_wrapper.py:33-50)
DIRECTIONS = [1, 3, 2, 4, 12, 32, 14, 34, 112, 332, 114, 334, 212, 232, 414, 434]

# name: (lower, upper, default)
SHARED = {
    "gravity": (-1000, -1e-6, -9.8), "friction": (0, 100, 1), "elasticity": (0, 100, 0),
    "ang_damping": (-INF, INF, -0.05), "viscosity": (0, INF, 0),
    "target_distance": (0, INF, 100), "target_radius": (0.1, INF, 5),
    "goal_position_x": (0, INF, 0.45), "goal_position_y": (0, INF, 0.05), "goal_position_z": (0, INF, 0.05),
    "joint_stiffness": (0.01, 100, 1.0),
}
MASS_BOUNDS = (1e-6, INF)
MACROS = {"GOAL": ("target_distance", "target_direction", "target_radius"),
          "GOAL_POSITION": ("goal_position_x", "goal_position_y", "goal_position_z")}

_HUMANOID_LINKS = ("mass_torso=10 mass_lwaist=2.2619467 mass_pelvis=6.6161942 mass_right_thigh=4.751751 "
                   "mass_right_shin=4.522842 mass_left_thigh=4.751751 mass_left_shin=4.522842 "
                   "mass_right_upper_arm=1.6610805 mass_right_lower_arm=1.2295402 mass_left_upper_arm=1.6610805 "
                   "mass_left_lower_arm=1.2295402")
_PHYSICS = "gravity friction elasticity ang_damping viscosity"

COLUMNS = {
    "ant": "gravity friction elasticity ang_damping mass_torso=10 viscosity GOAL",
    "halfcheetah": f"{_PHYSICS} mass_torso=10 mass_bthigh=1.5435146 mass_bshin=1.5874476 mass_bfoot=1.0953975 "
                   "mass_fthigh=1.4380753 mass_fshin=1.2008368 mass_ffoot=0.8845188 GOAL",
    "humanoid": f"{_PHYSICS} {_HUMANOID_LINKS} GOAL",
    "humanoidstandup": f"{_PHYSICS} {_HUMANOID_LINKS}",
    "hopper": f"{_PHYSICS} mass_torso=10 mass_thigh=4.0578904 mass_leg=2.7813568 mass_foot=5.3155746 GOAL",
    "walker2d": f"{_PHYSICS} mass_torso=10 mass_thigh=4.0578904 mass_leg=2.7813568 mass_foot=3.1667254 "
                "mass_thigh_left=4.0578904 mass_leg_left=2.7813568 mass_foot_left=3.1667254 GOAL",
    "inverted_pendulum": "gravity friction elasticity mass_cart=1 mass_pole=1 ang_damping viscosity",
    "inverted_double_pendulum": "gravity friction elasticity mass_cart=1 mass_pole=1 mass_pole2=1 ang_damping viscosity",
    "reacher": f"{_PHYSICS} mass_body0=0.03560472 mass_body1=0.03979351",
    "pusher": f"{_PHYSICS} mass_r_shoulder_pan_link=7.2935214 mass_r_shoulder_lift_link={math.pi!r} "
              "mass_r_upper_arm_roll_link=1.7140529 mass_r_elbow_flex_link=0.40715042 "
              "mass_r_forearm_roll_link=0.92818356 mass_r_wrist_flex_link=0.0050265482 "
              "mass_r_wrist_roll_link=0.18346901 mass_object=0.0018325957 GOAL_POSITION",
}


# This build's extension columns (never part of the DEFAULT tables, so that the default context, the "context"
# observation and the observation space have the reference's shape -- ADVICE r01): appended after the reference's
# columns by the opt-in classes CARLBraxHalfcheetahStiffness / CARLBraxHumanoidStiffness (BASELINE config 5).
EXTENSION_COLUMNS = {"halfcheetah": "joint_stiffness", "humanoid": "joint_stiffness"}


def _columns(family: str, extensions: bool = False) -> list[tuple[str, float | None]]:
    out: list[tuple[str, float | None]] = []
    spec = COLUMNS[family] + (" " + EXTENSION_COLUMNS.get(family, "") if extensions else "")
    for token in spec.split():
        if token in MACROS:
            out += [(name, None) for name in MACROS[token]]
        elif "=" in token:
            name, default = token.split("=")
            out.append((name, float(default)))
        else:
            out.append((token, None))
    return out


def masses(family: str) -> dict[str, float]:
    """``mass_<link>`` -> CARL default, in column order (the nominal values ``models._wire_context`` scales by)"""
    return {name: default for name, default in _columns(family) if default is not None}


def feature_table(family: str, extensions: bool = False) -> dict[str, ContextFeature]:
    feats: dict[str, ContextFeature] = {}
    for name, mass in _columns(family, extensions):
        if name == "target_direction":
            feats[name] = CategoricalContextFeature(name, choices=DIRECTIONS, default_value=1)
            continue
        lower, upper, default = (*MASS_BOUNDS, mass) if mass is not None else SHARED[name]
        feats[name] = UniformFloatContextFeature(name, lower=lower, upper=upper, default_value=default)
    return feats


# Smallest context / default ratio of every link-mass feature for which the explicit (semi-implicit Euler) spring
# integration of this build's models stays finite under a full-range random policy -- measured on an MI355X by
# tools/mass_stability_sweep.py (150 env steps, ratios 0.1 .. 1.0; features not listed: stable down to 0.1), plus a
# 10 % margin.  A lighter effective mass raises k dt^2 / m of the stiff constraint springs past the integrator's
# bound and the env blows up within a few steps; ``CARLBraxEnv`` refuses such contexts instead of producing NaNs
# (the reference cannot reach this regime: its context update never reaches the physics, Quirk B1).
_HUMANOID_FLOORS = {"mass_torso": 0.31, "mass_lwaist": 0.21, "mass_pelvis": 0.31, "mass_right_thigh": 0.21,
                    "mass_left_thigh": 0.21, "mass_right_upper_arm": 0.21, "mass_left_upper_arm": 0.21}
MASS_RATIO_FLOOR = {
    "ant": {"mass_torso": 0.37},
    "halfcheetah": {"mass_torso": 0.80, "mass_bthigh": 0.63, "mass_bshin": 0.55, "mass_bfoot": 0.29,
                    "mass_fthigh": 0.61, "mass_fshin": 0.55, "mass_ffoot": 0.29},
    "humanoid": _HUMANOID_FLOORS, "humanoidstandup": _HUMANOID_FLOORS,
    "hopper": {"mass_torso": 0.13, "mass_thigh": 0.13, "mass_leg": 0.13},
    "walker2d": {"mass_torso": 0.19, "mass_thigh": 0.14, "mass_leg": 0.13, "mass_thigh_left": 0.14, "mass_leg_left": 0.13},
    "inverted_pendulum": {"mass_cart": 0.85, "mass_pole": 0.64},
    "inverted_double_pendulum": {"mass_cart": 0.35, "mass_pole": 0.37, "mass_pole2": 0.19},
    "reacher": {"mass_body0": 0.42, "mass_body1": 0.22},
    "pusher": {"mass_r_wrist_flex_link": 0.88, "mass_object": 0.22},
}
# ``joint_stiffness`` (this build's extension feature, BASELINE config 5) scales the constraint stiffness k of the spring backend;
# the explicit integration needs k dt^2 / m below its bound just as it does for a light link.  Measured ceilings of the
# SCALE at default masses (tools/stiffness_stability_sweep.py on an MI355X, round 6: 1 024 envs x 300 steps per cell under
# random actions -- Halfcheetah: none of 1 024 blows up at x 2.0 with mass_torso >= 0.8 x default, all at x 2.5 / 0.8 and
# x 3.0 / 1.0, all at x 4 whatever the torso; Humanoid: none at x 6, all at x 10 / 0.8).  BASELINE config 5 samples
# U(0.5, 2): inside.  ``CARLBraxEnv`` warns (mass_check="warn") or refuses ("error") above them; the physics is NOT clamped.
JOINT_STIFFNESS_CEILING = {"halfcheetah": 2.0, "humanoid": 6.0, "humanoidstandup": 6.0}
DEFAULT_MASS_RATIO_FLOOR = 0.1  # nothing below was measured
# Several light links at once are less stable than each alone: with EVERY mass feature at alpha x its floor the
# envs still blew up to the alpha below (tools/mass_combo_sweep.py on an MI355X, round 3: profiles/
# r03_mass_combo_sweep.txt) -- the floors of an env with two or more light links are these multiples (+10 %) of the
# single-feature ones, capped at the default mass.
COMBINED_FLOOR_SCALE = {"ant": 1.1, "halfcheetah": 1.45, "humanoid": 1.85, "humanoidstandup": 1.85, "hopper": 1.65,
                        "walker2d": 1.65, "inverted_pendulum": 1.2, "inverted_double_pendulum": 1.6, "reacher": 1.45,
                        "pusher": 1.15}
