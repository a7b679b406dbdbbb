"""Device context sampler: spec construction from CARL feature objects and
the CPU restatement of the kernel (oracle/context_sampler.c).  The device stream is a new
(Philox) stream, so what is checked here is the distribution family per feature type, default
fill, determinism and shard invariance -- the reference's exact NumPy draws are pinned for the
host sampler in tests/test_context_sampler.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from carl_amd import _lib
from carl_amd.context.context_space import (
    CategoricalContextFeature,
    ContextSpace,
    NormalFloatContextFeature,
    UniformFloatContextFeature,
    UniformIntegerContextFeature,
)
from carl_amd.context.device_sampler import build_specs
from oracle import oracle as O

SPACE = ContextSpace({
    "gravity": UniformFloatContextFeature("gravity", lower=0.1, upper=np.inf, default_value=9.8),
    "length": UniformFloatContextFeature("length", lower=0.05, upper=5.0, default_value=0.5),
    "mass": UniformFloatContextFeature("mass", lower=1e-3, upper=10.0, default_value=1.0),
    "n_legs": UniformIntegerContextFeature("n_legs", lower=1, upper=8, default_value=4),
    "direction": CategoricalContextFeature("direction", choices=[1, 3, 2, 4, 12, 32], default_value=1),
    "noise": UniformFloatContextFeature("noise", lower=-np.inf, upper=np.inf, default_value=0.0),
})
DISTS = [
    UniformFloatContextFeature("gravity", 5, 15),
    NormalFloatContextFeature("length", mu=0.5, sigma=0.4, lower=0.05, upper=5.0),
    UniformFloatContextFeature("mass", 0.01, 10.0, log=True),
    UniformIntegerContextFeature("n_legs", 2, 6),
    CategoricalContextFeature("direction", choices=[1, 3, 2, 4, 12, 32]),
]


def test_feature_spec_layout_matches_c(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f[0] for f in _lib.FeatureSpec._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{root}/include/carl_amd.h"', "int main(void){",
           'printf("%zu\\n", sizeof(carl_feature_spec_t));']
    src += [f'printf("%zu\\n", offsetof(carl_feature_spec_t, {f}));' for f in fields] + ["return 0;}"]
    (tmp_path / "l.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-o", str(tmp_path / "l"), str(tmp_path / "l.c")], check=True)
    out = list(map(int, subprocess.run([str(tmp_path / "l")], capture_output=True, text=True, check=True).stdout.split()))
    assert out[0] == C.sizeof(_lib.FeatureSpec)
    assert out[1:] == [getattr(_lib.FeatureSpec, f).offset for f in fields]


def test_specs_follow_space_order_and_defaults():
    names, specs = build_specs(SPACE, DISTS)
    assert names == ["gravity", "length", "mass", "n_legs", "direction", "noise"]
    kinds = [specs[j].kind for j in range(6)]
    assert kinds == [_lib.FEAT_UNIFORM_FLOAT, _lib.FEAT_NORMAL_FLOAT, _lib.FEAT_UNIFORM_FLOAT, _lib.FEAT_UNIFORM_INT,
                     _lib.FEAT_CATEGORICAL, _lib.FEAT_CONSTANT]
    assert specs[2].log_scale == 1 and specs[5].value == 0.0 and specs[4].n_choices == 6
    with pytest.raises(ValueError):  # same failure class as the reference's unknown-feature check
        build_specs(SPACE, [UniformFloatContextFeature("nonexistent", 0, 1)])
    with pytest.raises(ValueError):
        build_specs(SPACE, [CategoricalContextFeature("direction", choices=["north", "south"])])


def test_distributions_of_the_restated_sampler():
    _, specs = build_specs(SPACE, DISTS)
    n = 200_000
    t = O.sample_contexts(specs, n, seed=123).astype(np.float64)
    g, length, mass, legs, direction, noise = t
    assert g.min() >= 5 and g.max() <= 15 and abs(g.mean() - 10) < 0.03 and abs(g.var() - 100 / 12) < 0.1
    # bounded normal: redraws outside [0.05, 5] -> compare with the truncated-normal moments
    from scipy import stats

    tn = stats.truncnorm((0.05 - 0.5) / 0.4, (5.0 - 0.5) / 0.4, loc=0.5, scale=0.4)
    assert length.min() >= 0.05 and length.max() <= 5.0
    assert abs(length.mean() - tn.mean()) < 4e-3 and abs(length.std() - tn.std()) < 4e-3
    assert stats.kstest(length[:20000], tn.cdf).pvalue > 1e-3
    # log-uniform: log(mass) uniform on [log .01, log 10]
    lm = np.log(mass)
    assert mass.min() >= 0.01 * (1 - 1e-6) and mass.max() <= 10.0 * (1 + 1e-6)
    assert abs(lm.mean() - 0.5 * (np.log(0.01) + np.log(10))) < 0.02
    assert set(np.unique(legs)) == {2.0, 3.0, 4.0, 5.0, 6.0}
    assert np.abs(np.bincount(legs.astype(int))[2:] / n - 0.2).max() < 5e-3
    assert set(np.unique(direction)) == {1.0, 3.0, 2.0, 4.0, 12.0, 32.0}
    assert np.abs(np.array([(direction == c).mean() for c in (1, 3, 2, 4, 12, 32)]) - 1 / 6).max() < 5e-3
    assert np.all(noise == 0.0)  # not sampled: default
    # columns are independent streams
    assert abs(np.corrcoef(g, length)[0, 1]) < 0.01 and abs(np.corrcoef(g, np.log(mass))[0, 1]) < 0.01
    assert O.verify_contexts(specs, t) == 0


def test_determinism_seed_and_shard_invariance():
    _, specs = build_specs(SPACE, DISTS)
    a = O.sample_contexts(specs, 4096, seed=7)
    b = O.sample_contexts(specs, 4096, seed=7)
    c = O.sample_contexts(specs, 4096, seed=8)
    assert np.array_equal(a, b) and not np.array_equal(a[0], c[0])
    lo = O.sample_contexts(specs, 1024, seed=7, context_offset=0)
    hi = O.sample_contexts(specs, 3072, seed=7, context_offset=1024)
    assert np.array_equal(np.concatenate([lo, hi], axis=1), a)  # a rank's shard = its slice of the global set


def test_verify_counts_out_of_bounds_entries():
    names, specs = build_specs(SPACE, DISTS)
    t = O.sample_contexts(specs, 512, seed=1)
    t[0, 3] = 1e9        # gravity above the sampled range's upper bound (15)
    t[1, 10] = -1.0      # length below its lower bound
    t[4, 7] = 5.0        # not a direction code
    t[3, 0] = np.nan
    assert O.verify_contexts(specs, t) == 4
