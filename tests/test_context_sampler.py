"""ContextSampler: the reference's test (test/test_context_sampler.py:33-62) plus the
only pinned numbers in the reference tree -- the recorded notebook outputs
(tests/golden/notebook_sampler_outputs.json).  CPU-only."""
import json
import os

import numpy as np
import pytest

from carl_amd.context.context_space import (
    CategoricalContextFeature,
    ContextSpace,
    NormalFloatContextFeature,
    UniformFloatContextFeature,
)
from carl_amd.context.sampler import ContextSampler

context_space_dict = {"gravity": UniformFloatContextFeature("gravity", lower=1, upper=10, default_value=9.8)}
sample_dist = {"gravity": NormalFloatContextFeature("gravity", mu=9.8, sigma=0.0, default_value=9.8, upper=20, lower=1)}

# CARLBraxAnt's feature table (carl/envs/brax/carl_ant.py:20-49), needed for the notebook cases
DIRECTIONS = [1, 3, 2, 4, 12, 32, 14, 34, 112, 332, 114, 334, 212, 232, 414, 434]
U = UniformFloatContextFeature
ANT_SPACE = ContextSpace({
    "gravity": U("gravity", lower=-1000, upper=-1e-6, default_value=-9.8),
    "friction": U("friction", lower=0, upper=100, default_value=1),
    "elasticity": U("elasticity", lower=0, upper=100, default_value=0),
    "ang_damping": U("ang_damping", lower=-np.inf, upper=np.inf, default_value=-0.05),
    "mass_torso": U("mass_torso", lower=1e-6, upper=np.inf, default_value=10),
    "viscosity": U("viscosity", lower=0, upper=np.inf, default_value=0),
    "target_distance": U("target_distance", lower=0, upper=np.inf, default_value=100),
    "target_direction": CategoricalContextFeature("target_direction", choices=DIRECTIONS, default_value=1),
    "target_radius": U("target_radius", lower=0.1, upper=np.inf, default_value=5),
})


@pytest.fixture(scope="module")
def recorded(golden_dir):
    return json.load(open(os.path.join(golden_dir, "notebook_sampler_outputs.json")))


def test_init_forms():
    cspace = ContextSpace(context_space_dict)
    ContextSampler(context_distributions=sample_dist, context_space=cspace, seed=0, name="TestSampler")
    ContextSampler(context_distributions=list(sample_dist.values()), context_space=cspace, seed=0)
    with pytest.raises(ValueError):
        ContextSampler(context_distributions=0, context_space=cspace, seed=0, name="TestSampler")


def test_sample_contexts_sigma0():
    sampler = ContextSampler(sample_dist, ContextSpace(context_space_dict), seed=0, name="TestSampler")
    contexts = sampler.sample_contexts(n_contexts=3)
    assert len(contexts) == 3 and contexts[0]["gravity"] == 9.8
    contexts = sampler.sample_contexts(n_contexts=1)
    assert len(contexts) == 1 and contexts[0]["gravity"] == 9.8


def test_recorded_single_normal(recorded):
    """examples/sample_contexts_with_brax.ipynb cell 5: exact float equality"""
    r = recorded["single_normal"]
    f = r["feature"]
    s = ContextSampler([NormalFloatContextFeature(f["name"], mu=f["mu"], sigma=f["sigma"], upper=f["upper"],
                                                  lower=f["lower"])], ANT_SPACE, seed=r["seed"])
    contexts = s.sample_contexts(r["n"])
    assert [contexts[i]["gravity"] for i in range(r["n"])] == r["gravity"]
    want_rest = {k: v for k, v in recorded["ant_default_context"].items() if k != "gravity"}
    assert {k: v for k, v in contexts[3].items() if k != "gravity"} == want_rest
    assert list(contexts[0].keys()) == list(recorded["ant_default_context"].keys())


def test_recorded_normal_plus_categorical(recorded):
    """examples/brax_with_goals.ipynb cell 1: pins name-sorted per-feature vector draws"""
    r = recorded["normal_plus_categorical"]
    s = ContextSampler([NormalFloatContextFeature("target_distance", mu=9.8, sigma=1, upper=50, lower=0),
                        CategoricalContextFeature("target_direction", choices=r["choices"])], ANT_SPACE, seed=r["seed"])
    contexts = s.sample_contexts(r["n"])
    assert [contexts[i]["target_distance"] for i in range(r["n"])] == r["target_distance"]
    assert [contexts[i]["target_direction"] for i in range(r["n"])] == r["target_direction"]


def test_recorded_two_normals(recorded):
    """examples/brax_with_goals.ipynb cell 4 (first three contexts are visible in the output)"""
    r = recorded["two_normals"]
    space = ContextSpace({"goal_position_x": U("goal_position_x", lower=-np.inf, upper=np.inf, default_value=0.45),
                          "goal_position_y": U("goal_position_y", lower=-np.inf, upper=np.inf, default_value=-0.05)})
    s = ContextSampler([NormalFloatContextFeature("goal_position_x", mu=9.8, sigma=1, upper=50, lower=0),
                        NormalFloatContextFeature("goal_position_y", mu=9.8, sigma=1, upper=50, lower=0)], space,
                       seed=r["seed"])
    contexts = s.sample_contexts(r["n"])
    assert [contexts[i]["goal_position_x"] for i in range(3)] == r["goal_position_x"]
    assert [contexts[i]["goal_position_y"] for i in range(3)] == r["goal_position_y"]


def test_table_path_equals_dict_path():
    """the dense form consumed by the engine holds exactly the dict path's draws"""
    dists = [UniformFloatContextFeature("g", 1, 20), UniformFloatContextFeature("l", 0.5, 2.0)]
    from carl_amd.context.context_space import ContextSpace as CS

    names = ["gravity", "dt", "g", "m", "l", "initial_angle_max", "initial_velocity_max"]
    defaults = [8.0, 0.05, 10, 1, 1, np.pi, 1]
    space = CS({n: U(n, lower=-np.inf, upper=np.inf, default_value=d) for n, d in zip(names, defaults)})
    a = ContextSampler(dists, space, seed=0).sample_contexts(200)
    t = ContextSampler(dists, space, seed=0).sample_context_table(200)
    assert t.names == names and len(t) == 200
    for i in (0, 17, 199):
        assert t[i] == {k: float(v) for k, v in a[i].items()}
    g = t.column("g")
    assert g.min() >= 1 and g.max() <= 20 and np.unique(g).size == 200
    assert (t.column("m") == 1).all()


def test_normal_bounds_are_honoured():
    f = NormalFloatContextFeature("x", mu=0.0, sigma=5.0, lower=-1.0, upper=1.0)
    v = f.rvs(size=2000, random_state=1)
    assert v.min() >= -1.0 and v.max() <= 1.0


def test_log_scale_features_sample_on_the_log_scale():
    """ConfigSpace's log=True (search-space JSON "log": true): uniform on the LOG scale, default = the centre of
    the range on the log scale.  (the host sampler used to ignore the flag -- median ~501 and default
    500.0005 for U(1e-3, 1e3, log=True) -- while the device sampler honoured it.)"""
    import math

    from carl_amd.context.features import (
        NormalFloatContextFeature,
        UniformFloatContextFeature,
        UniformIntegerContextFeature,
    )

    f = UniformFloatContextFeature("a", 1e-3, 1e3, log=True)
    assert f.default_value == pytest.approx(1.0)  # geometric mean
    v = f.rvs(size=20000, random_state=0)
    assert v.min() >= 1e-3 and v.max() <= 1e3
    assert 0.8 < np.median(v) < 1.25  # log-uniform: median = geometric mean (linear sampling gives ~500)
    lg = np.log10(v)
    assert abs(lg.mean()) < 0.05 and abs(lg.std() - 6 / math.sqrt(12)) < 0.05  # uniform over [-3, 3] decades
    # same stream as the linear feature: the log flag only changes the map from the uniform draw
    lin = UniformFloatContextFeature("a", 1e-3, 1e3).rvs(size=5, random_state=3)
    u = (lin - 1e-3) / (1e3 - 1e-3)
    np.testing.assert_allclose(f.rvs(size=5, random_state=3), np.exp(math.log(1e-3) + u * math.log(1e6)), rtol=1e-12)

    g = UniformIntegerContextFeature("n", 1, 1000, log=True)
    assert g.default_value == 32  # round(sqrt(1 * 1000))
    w = g.rvs(size=20000, random_state=1)
    assert w.min() >= 1 and w.max() <= 1000 and w.dtype.kind == "i"
    assert 20 < np.median(w) < 45  # linear sampling gives ~500
    with pytest.raises(ValueError):
        UniformFloatContextFeature("b", 0.0, 1.0, log=True)  # log scale needs lower > 0
    with pytest.raises(NotImplementedError):
        NormalFloatContextFeature("c", mu=1.0, sigma=1.0, lower=0.1, upper=10, log=True)


def test_log_scale_feature_through_the_sampler_and_search_space():
    from carl_amd.context.context_space import ContextSpace, UniformFloatContextFeature
    from carl_amd.context.sampler import ContextSampler
    from carl_amd.context.search_space_encoding import search_space_to_config_space

    space = ContextSpace({"k": UniformFloatContextFeature("k", 1e-2, 1e2, default_value=1.0)})
    s = ContextSampler([UniformFloatContextFeature("k", 1e-2, 1e2, log=True)], space, seed=0)
    t = s.sample_context_table(4000)
    k = t.values_2d[:, t.names.index("k")]
    assert 0.8 < np.median(k) < 1.25
    cs = search_space_to_config_space(
        {"hyperparameters": [{"name": "k", "type": "uniform_float", "log": True, "lower": 0.01, "upper": 100.0,
                              "default": 1.0}]})
    assert cs["k"].log is True
    v = cs["k"].rvs(size=4000, random_state=0)
    assert 0.8 < np.median(v) < 1.25
