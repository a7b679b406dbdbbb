"""Context-feature types and a small seeded configuration space.

The reference aliases ConfigSpace hyperparameters as its "context features"
(carl/context/context_space.py:23-28) and subclasses ``ConfigurationSpace`` for its
sampler (carl/context/sampler.py:11).  ConfigSpace is not part of this build, so the
four feature kinds CARL uses are provided here with the attribute surface CARL
touches: ``name, lower, upper, default_value, choices, mu, sigma, rvs()``.

Sampling order is pinned by the recorded outputs in the reference's notebooks
(examples/sample_contexts_with_brax.ipynb cell 5, examples/brax_with_goals.ipynb
cells 1 and 4): hyperparameters are visited in NAME-SORTED
order, each draws its whole size-n vector from one legacy ``RandomState(seed)``:
categorical -> ``choice(n_choices, size, p=uniform)``, normal -> ``normal(mu, sigma,
size)`` (bounds honoured by redrawing out-of-range entries), uniform float ->
``lower + (upper - lower) * uniform(size)``.
"""
from __future__ import annotations

import math
from typing import Any, Iterable, Sequence

import numpy as np


class ContextFeature:
    """Base class (ConfigSpace ``Hyperparameter`` stand-in)."""

    name: str
    default_value: Any

    def rvs(self, size: int | None = None, random_state: Any = None):
        rs = _as_random_state(random_state)
        v = self._sample_vector(1 if size is None else int(size), rs)
        if size is None:
            return self._to_python(v[0])
        return v

    def _sample_vector(self, size: int, rs: np.random.RandomState) -> np.ndarray:
        raise NotImplementedError

    @staticmethod
    def _to_python(v):
        return v.item() if hasattr(v, "item") else v

    def is_legal(self, value: Any) -> bool:
        return True


def _as_random_state(random_state) -> np.random.RandomState:
    if random_state is None:
        return np.random.mtrand._rand  # the global legacy state, like scipy-style rvs()
    if isinstance(random_state, np.random.RandomState):
        return random_state
    return np.random.RandomState(int(random_state))


class NumericalContextFeature(ContextFeature):
    def __init__(self, name: str, lower: float, upper: float, default_value: float | None = None,
                 log: bool = False, meta: dict | None = None):
        if lower is not None and upper is not None and lower > upper:
            raise ValueError(f"{name}: lower bound {lower} exceeds upper bound {upper}")
        if log and not (lower is not None and lower > 0):
            raise ValueError(f"{name}: log=True needs a positive lower bound, got {lower}")
        self.name = name
        self.lower = lower
        self.upper = upper
        self.log = log
        self.meta = meta
        self.default_value = self._check_default(default_value)

    def _check_default(self, default_value):
        if default_value is None:
            lo, hi = self.lower, self.upper
            if lo is None or hi is None or math.isinf(lo) or math.isinf(hi):
                return 0.0
            if self.log:  # ConfigSpace: the centre of the range ON THE LOG SCALE (geometric mean)
                return math.exp((math.log(lo) + math.log(hi)) / 2)
            return (lo + hi) / 2
        if not self.is_legal(default_value):
            raise ValueError(
                f"{self.name}: default value {default_value} outside [{self.lower}, {self.upper}]"
            )
        return default_value

    def is_legal(self, value) -> bool:
        lo = -math.inf if self.lower is None else self.lower
        hi = math.inf if self.upper is None else self.upper
        return bool(lo <= value <= hi)

    def __repr__(self):
        return (f"{type(self).__name__}({self.name!r}, lower={self.lower}, upper={self.upper}, "
                f"default_value={self.default_value})")


class UniformFloatContextFeature(NumericalContextFeature):
    def __init__(self, name: str, lower: float, upper: float, default_value: float | None = None,
                 log: bool = False, meta: dict | None = None):
        super().__init__(name, float(lower), float(upper), default_value, log, meta)
        if self.default_value is not None:
            self.default_value = float(self.default_value)

    def _sample_vector(self, size, rs):
        u = rs.uniform(size=size)
        if self.log:  # log-uniform: uniform on the log scale, as ConfigSpace's log=True (and the device sampler)
            llo, lhi = math.log(self.lower), math.log(self.upper)
            return np.clip(np.exp(llo + (lhi - llo) * u), self.lower, self.upper)
        with np.errstate(invalid="ignore", over="ignore"):
            return self.lower + (self.upper - self.lower) * u


class NormalFloatContextFeature(NumericalContextFeature):
    def __init__(self, name: str, mu: float, sigma: float, lower: float | None = None,
                 upper: float | None = None, default_value: float | None = None,
                 log: bool = False, meta: dict | None = None):
        self.mu = float(mu)
        self.sigma = float(sigma)
        lo = -math.inf if lower is None else float(lower)
        hi = math.inf if upper is None else float(upper)
        if log:
            # ConfigSpace's log-normal parametrisation is version-dependent (mu / sigma on which scale), the
            # reference never uses it and the device sampler has no such kind: refuse instead of guessing
            raise NotImplementedError(f"{name}: NormalFloatContextFeature(log=True) is not supported")
        super().__init__(name, lo, hi, self.mu if default_value is None else default_value, log, meta)
        self.default_value = float(self.default_value)

    def _sample_vector(self, size, rs):
        v = rs.normal(self.mu, self.sigma, size)
        bad = (v < self.lower) | (v > self.upper)
        tries = 0
        while bad.any() and tries < 1000:
            v[bad] = rs.normal(self.mu, self.sigma, int(bad.sum()))
            bad = (v < self.lower) | (v > self.upper)
            tries += 1
        return np.clip(v, self.lower, self.upper)


class UniformIntegerContextFeature(NumericalContextFeature):
    def __init__(self, name: str, lower: int, upper: int, default_value: int | None = None,
                 log: bool = False, meta: dict | None = None):
        super().__init__(name, int(lower), int(upper), default_value, log, meta)
        self.default_value = int(round(self.default_value))

    def _sample_vector(self, size, rs):
        if self.log:  # uniform on the log scale over [lower - 1/2, upper + 1/2), rounded (ConfigSpace's rule)
            llo, lhi = math.log(self.lower - 0.49999), math.log(self.upper + 0.49999)
            v = np.rint(np.exp(llo + (lhi - llo) * rs.uniform(size=size)))
            return np.clip(v, self.lower, self.upper).astype(np.int64)
        return rs.randint(self.lower, self.upper + 1, size=size)


class CategoricalContextFeature(ContextFeature):
    def __init__(self, name: str, choices: Sequence[Any], default_value: Any = None,
                 weights: Sequence[float] | None = None, meta: dict | None = None):
        choices = list(choices)
        if len(choices) == 0:
            raise ValueError(f"{name}: categorical feature needs at least one choice")
        if len(set(map(repr, choices))) != len(choices):
            raise ValueError(f"{name}: duplicate choices")
        self.name = name
        self.choices = tuple(choices)
        self.num_choices = len(choices)
        if weights is None:
            self.probabilities = np.full(len(choices), 1.0 / len(choices))
        else:
            w = np.asarray(weights, dtype=np.float64)
            self.probabilities = w / w.sum()
        if default_value is None:
            default_value = choices[0]
        if default_value not in choices:
            raise ValueError(f"{name}: default {default_value!r} is not one of the choices")
        self.default_value = default_value
        self.meta = meta

    def is_legal(self, value) -> bool:
        return value in self.choices

    def _sample_vector(self, size, rs):
        idx = rs.choice(self.num_choices, size=size, replace=True, p=self.probabilities)
        out = np.empty(size, dtype=object)
        for i, j in enumerate(idx):
            out[i] = self.choices[int(j)]
        try:
            return np.array(out.tolist())
        except Exception:  # pragma: no cover - ragged choices
            return out

    def __repr__(self):
        return f"CategoricalContextFeature({self.name!r}, choices={self.choices}, default_value={self.default_value!r})"


class Configuration(dict):
    """One sampled configuration; behaves like the dict CARL converts it to."""


class ConfigurationSpace:
    """Seeded, ordered container of features (the slice of ConfigSpace CARL uses:
    ``add_hyperparameters``, ``values``, ``sample_configuration``, ``seed``)."""

    def __init__(self, name: str | None = None, seed: int | None = None, space: dict | None = None):
        self.name = name
        self._hps: dict[str, ContextFeature] = {}
        self.random = np.random.RandomState(seed)
        if space:
            self.add_hyperparameters(_features_from_shorthand(space))

    def seed(self, seed: int | None = None) -> None:
        self.random = np.random.RandomState(seed)

    def add_hyperparameters(self, hyperparameters: Iterable[ContextFeature]) -> None:
        for hp in hyperparameters:
            if not isinstance(hp, ContextFeature):
                raise TypeError(f"Expected a context feature, got {type(hp)}")
            if hp.name in self._hps:
                raise ValueError(f"Feature {hp.name!r} is already in the space")
            self._hps[hp.name] = hp
        # name-sorted iteration order (pinned by the recorded notebook outputs)
        self._hps = dict(sorted(self._hps.items(), key=lambda kv: kv[0]))

    add = add_hyperparameters

    def get_hyperparameters(self) -> list[ContextFeature]:
        return list(self._hps.values())

    def values(self):
        return self._hps.values()

    def keys(self):
        return self._hps.keys()

    def items(self):
        return self._hps.items()

    def __getitem__(self, name: str) -> ContextFeature:
        return self._hps[name]

    def __contains__(self, name) -> bool:
        return name in self._hps

    def __len__(self) -> int:
        return len(self._hps)

    def __iter__(self):
        return iter(self._hps)

    def sample_vectors(self, size: int) -> dict[str, np.ndarray]:
        """Column-wise draw: one size-``size`` vector per feature, name-sorted."""
        return {name: hp._sample_vector(size, self.random) for name, hp in self._hps.items()}

    def sample_configuration(self, size: int = 1):
        cols = self.sample_vectors(size)
        rows = [Configuration({k: ContextFeature._to_python(v[i]) for k, v in cols.items()})
                for i in range(size)]
        return rows[0] if size == 1 else rows


def _features_from_shorthand(space: dict) -> list[ContextFeature]:
    """ConfigSpace's ``space={...}`` shorthand: (int,int) -> uniform integer,
    (float,float) -> uniform float, list -> categorical, scalar -> constant."""
    feats: list[ContextFeature] = []
    for name, spec in space.items():
        if isinstance(spec, ContextFeature):
            feats.append(spec)
        elif isinstance(spec, tuple) and len(spec) == 2:
            lo, hi = spec
            if isinstance(lo, int) and isinstance(hi, int):
                feats.append(UniformIntegerContextFeature(name, lo, hi))
            else:
                feats.append(UniformFloatContextFeature(name, lo, hi))
        elif isinstance(spec, (list, tuple)):
            feats.append(CategoricalContextFeature(name, list(spec)))
        else:
            feats.append(CategoricalContextFeature(name, [spec], default_value=spec))
    return feats
