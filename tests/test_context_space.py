"""Context space / bounds / search-space encoding: the reference's own assertions
(test/test_context_space.py:43-109, test/test_context_bounds.py:8-59,
test/test_search_space_encoding.py:48-77) on this build's types.  CPU-only."""
import json

import numpy as np
import pytest

from carl_amd import spaces
from carl_amd.context.context_space import (
    ContextSpace,
    NormalFloatContextFeature,
    UniformFloatContextFeature,
    UniformIntegerContextFeature,
)
from carl_amd.context.features import ConfigurationSpace
from carl_amd.context.search_space_encoding import search_space_to_config_space
from carl_amd.context.table import ContextTable
from carl_amd.context.utils import get_context_bounds

U = UniformFloatContextFeature
context_space_dict = {
    "gravity": U("gravity", lower=0.1, upper=np.inf, default_value=9.8),
    "masscart": U("masscart", lower=0.1, upper=10, default_value=1.0),
    "masspole": U("masspole", lower=0.01, upper=1, default_value=0.1),
    "length": U("length", lower=0.05, upper=5, default_value=0.5),
    "force_mag": U("force_mag", lower=1, upper=100, default_value=10.0),
    "tau": U("tau", lower=0.002, upper=0.2, default_value=0.02),
}
DEFAULT = {"gravity": 9.8, "masscart": 1, "masspole": 0.1, "length": 0.5, "force_mag": 10, "tau": 0.02}


@pytest.fixture
def cs():
    return ContextSpace(context_space=context_space_dict)


def test_insert_defaults(cs):
    assert cs.insert_defaults({}) == DEFAULT
    assert cs.insert_defaults({"tau": 0.1})["tau"] == 0.1
    # Quirk S1: keys outside context_keys still come through from `context`
    assert cs.insert_defaults({"tau": 0.1}, ["gravity"]) == {"gravity": 9.8, "tau": 0.1}


def test_get_default_context(cs):
    assert cs.get_default_context() == DEFAULT


def test_get_lower_and_upper_bound(cs):
    assert cs.get_lower_and_upper_bound("length") == (0.05, 5)


def test_to_gymnasium_space_type(cs):
    assert type(cs.to_gymnasium_space(as_dict=False)) is spaces.Box
    d = cs.to_gymnasium_space(as_dict=True)
    assert type(d) is spaces.Dict and list(d.spaces.keys()) == list(DEFAULT)
    box = cs.to_gymnasium_space(["length", "tau"])
    assert box.shape == (2,) and box.dtype == np.float32
    np.testing.assert_allclose(box.low, [0.05, 0.002], rtol=1e-6)


def test_to_gymnasium_space_other_types():
    c = ContextSpace({
        "gravity": U("gravity", lower=0.1, upper=np.inf, default_value=9.8),
        "masscart": UniformIntegerContextFeature("masscart", lower=1, upper=10, default_value=1),
    })
    c.to_gymnasium_space()


def test_verify_context(cs):
    assert cs.verify_context({"hihi": 39, "gravity": 3}) is False
    assert cs.verify_context({"masscart": -10}) is False
    assert cs.verify_context({"masscart": 2.0, "gravity": 3}) is True


def test_sample(cs):
    # infinite upper bound on gravity: rvs() of such a feature is not finite; use a bounded space
    bounded = ContextSpace({k: v for k, v in context_space_dict.items() if k != "gravity"})
    ctx = bounded.sample_contexts(["masscart"], size=1)
    assert bounded.verify_context(ctx)
    ctxs = bounded.sample_contexts(["masscart"], size=10)
    assert len(ctxs) == 10 and all(bounded.verify_context(c) for c in ctxs)
    ctxs = bounded.sample_contexts(None, size=10)
    assert len(ctxs) == 10 and all(bounded.verify_context(c) for c in ctxs)
    with pytest.raises(ValueError):
        cs.sample_contexts(["false_feature"], size=0)


def test_context_bounds():
    bounds = {"a": (-np.inf, np.inf, float), "b": (0, np.inf, float), "c": (-1.0, 2.5, float)}
    lo, hi = get_context_bounds(["c", "a", "b"], bounds)
    np.testing.assert_array_equal(lo, [-1.0, -np.inf, 0.0])
    np.testing.assert_array_equal(hi, [2.5, np.inf, np.inf])


def test_feature_validation():
    with pytest.raises(ValueError):
        U("x", lower=2, upper=1)
    with pytest.raises(ValueError):
        U("x", lower=0, upper=1, default_value=3)
    f = NormalFloatContextFeature("g", mu=9.8, sigma=0.0, default_value=9.8, upper=20, lower=1)
    assert f.rvs(random_state=0) == 9.8 and f.default_value == 9.8


def test_search_space_forms(tmp_path):
    space = ConfigurationSpace(name="myspace", space={
        "uniform_integer": (1, 10), "uniform_float": (1.0, 10.0), "categorical": ["a", "b", "c"], "constant": 1337})
    assert search_space_to_config_space(space) is space
    assert len(search_space_to_config_space({"hyperparameters": {}})) == 0
    d2 = {"hyperparameters": [
        {"name": "x0", "type": "uniform_float", "log": False, "lower": -512.0, "upper": 512.0, "default": -3.0, "q": None},
        {"name": "x1", "type": "uniform_float", "log": False, "lower": -512.0, "upper": 512.0, "default": -4.0, "q": None}],
        "conditions": [], "forbiddens": [], "python_module_version": "0.4.17", "json_format_version": 0.2}
    cs2 = search_space_to_config_space(d2, seed=3)
    assert list(cs2.keys()) == ["x0", "x1"] and cs2["x1"].default_value == -4.0
    p = tmp_path / "space.json"
    p.write_text(json.dumps(d2))
    assert list(search_space_to_config_space(str(p)).keys()) == ["x0", "x1"]
    hydra = {"hyperparameters": {"g": {"type": "normal_float", "mu": 9.8, "sigma": 1.0, "lower": 0, "upper": 50}}}
    assert search_space_to_config_space(hydra)["g"].mu == 9.8
    with pytest.raises(ValueError):
        search_space_to_config_space(0)


def test_context_table_roundtrip(cs):
    contexts = {"a": {"gravity": 5.0}, "b": {"tau": 0.1, "length": 1.0}, 7: {}}
    t = cs.to_table(contexts)
    assert t.names == list(DEFAULT) and len(t) == 3 and list(t) == ["a", "b", 7]
    assert t["a"] == {**DEFAULT, "gravity": 5.0}
    assert t["b"]["tau"] == 0.1 and t["b"]["gravity"] == 9.8 and t[7] == {k: float(v) for k, v in DEFAULT.items()}
    assert dict(t.items())["b"]["length"] == 1.0
    np.testing.assert_array_equal(cs.verify_table(t), [True, True, True])
    bad = ContextTable(t.names, np.where(np.arange(6) == 1, -10.0, t.values_2d))
    np.testing.assert_array_equal(cs.verify_table(bad), [False, False, False])
    with pytest.raises(ValueError):
        cs.to_table({0: {"nope": 1.0}})
