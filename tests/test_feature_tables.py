"""Context-feature tables of every env class built here against the reference's (golden fixture made
by tests/golden/make_feature_table_golden.py from the reference's class files):
same features in the same order -- the order is the column order of the context table -- with the
same kind, bounds / choices and default."""
from __future__ import annotations

import json
import math
import os

import pytest

import carl_amd.envs as E

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "context_feature_tables.json")))
GOAL_WRAPPER = GOLDEN.pop("_goal_wrapper")


# the default tables hold the reference's features and nothing else; joint_stiffness (BASELINE config 5) lives in
# the opt-in classes CARLBraxHalfcheetahStiffness / CARLBraxHumanoidStiffness only
EXTENSIONS: dict = {}
# carl_inverted_double_pendulum.py:32-34: key "mass_pole2" constructed with name "mass_pole"
KNOWN_NAME_SLIPS = {("CARLBraxInvertedDoublePendulum", "mass_pole2")}


def _f(v):
    return {"inf": math.inf, "-inf": -math.inf}.get(v, v) if isinstance(v, str) else v


@pytest.mark.parametrize("cls_name", sorted(GOLDEN))
def test_feature_table_matches_reference(cls_name):
    cls = getattr(E, cls_name, None)
    assert cls is not None, f"{cls_name} is not built"
    mine = cls.get_context_features()
    want = GOLDEN[cls_name]
    keys = [w["key"] for w in want]
    # the reference's features first, in its order; then only this build's documented extensions
    assert list(mine)[: len(keys)] == keys
    assert set(list(mine)[len(keys):]) <= EXTENSIONS.get(cls_name, set())
    for w in want:
        f = mine[w["key"]]
        assert type(f).__name__ == w["kind"]
        # the feature's own name is its key here; the reference has one slip (KNOWN_NAME_SLIPS)
        assert f.name == w["key"] and (w["name"] == w["key"] or (cls_name, w["key"]) in KNOWN_NAME_SLIPS)
        assert float(f.default_value) == _f(w["default_value"]), w["name"]
        if "choices" in w:
            assert [float(c) for c in f.choices] == w["choices"]
        else:
            assert float(f.lower) == _f(w["lower"]) and float(f.upper) == _f(w["upper"]), w["name"]


def test_every_reference_env_class_of_the_path_is_built():
    assert len(GOLDEN) == 15  # 5 classic-control + 10 Brax classes


def test_goal_wrapper_names_user_code_imports():
    """``directions`` / ``DIRECTION_NAMES`` of carl/envs/brax/brax_walker_goal_wrapper.py:16-50"""
    from carl_amd.envs.brax.brax_walker_goal_wrapper import DIRECTION_NAMES, directions

    assert directions == GOAL_WRAPPER["directions"]
    assert {str(k): v for k, v in DIRECTION_NAMES.items()} == GOAL_WRAPPER["DIRECTION_NAMES"]


def test_language_goal_sentences_equal_the_reference_output():
    """``CARLBraxEnv.describe_goal`` against sentences produced by running the reference's
    ``BraxLanguageWrapper.get_goal_desc`` (brax_walker_goal_wrapper.py:169-181) on the same contexts"""
    from carl_amd.envs.brax.carl_brax_env import CARLBraxEnv

    assert len(GOAL_WRAPPER["sentences"]) >= 4
    for case in GOAL_WRAPPER["sentences"]:
        assert CARLBraxEnv.describe_goal(case["context"]) == case["sentence"]


def test_extension_classes_append_joint_stiffness_after_the_reference_columns():
    from carl_amd.envs.brax.feature_tables import feature_table

    for fam in ("halfcheetah", "humanoid"):
        base, ext = feature_table(fam), feature_table(fam, extensions=True)
        assert list(ext)[:-1] == list(base) and list(ext)[-1] == "joint_stiffness"
        assert "joint_stiffness" not in base and ext["joint_stiffness"].default_value == 1.0
    assert feature_table("ant", extensions=True).keys() == feature_table("ant").keys()
