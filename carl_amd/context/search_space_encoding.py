"""Search-space -> configuration space (reference: carl/context/search_space_encoding.py:46-144).

Accepts the forms the reference accepts -- a path to a ConfigSpace-JSON file, a
mapping with a ``hyperparameters`` entry (list of dicts, or hydra-style dict of
dicts), or an existing ``ConfigurationSpace`` -- and understands the JSON types
``uniform_float``, ``normal_float``, ``uniform_int``, ``categorical``, ``constant``.
omegaconf's ``DictConfig`` is a Mapping, so it is handled by the mapping branch.
"""
from __future__ import annotations

import json
from collections.abc import Mapping
from typing import Any

from carl_amd.context.features import (
    CategoricalContextFeature,
    ConfigurationSpace,
    ContextFeature,
    NormalFloatContextFeature,
    UniformFloatContextFeature,
    UniformIntegerContextFeature,
)


def _feature_from_json(cfg: Mapping) -> ContextFeature:
    kind = cfg.get("type")
    name = cfg["name"]
    default = cfg.get("default", cfg.get("default_value"))
    if kind == "uniform_float":
        return UniformFloatContextFeature(name, cfg["lower"], cfg["upper"], default, bool(cfg.get("log", False)))
    if kind == "normal_float":
        return NormalFloatContextFeature(name, cfg["mu"], cfg["sigma"], cfg.get("lower"), cfg.get("upper"), default,
                                         bool(cfg.get("log", False)))
    if kind in ("uniform_int", "uniform_integer"):
        return UniformIntegerContextFeature(name, cfg["lower"], cfg["upper"], default, bool(cfg.get("log", False)))
    if kind == "categorical":
        return CategoricalContextFeature(name, list(cfg["choices"]), default, cfg.get("weights"))
    if kind == "constant":
        return CategoricalContextFeature(name, [cfg["value"]], cfg["value"])
    raise ValueError(f"Unsupported hyperparameter type {kind!r} for {name!r}")


def search_space_to_config_space(search_space: Any, seed: int | None = None) -> ConfigurationSpace:
    if isinstance(search_space, ConfigurationSpace):
        cs = search_space
    else:
        if isinstance(search_space, str):
            with open(search_space, "r") as f:
                search_space = json.loads(f.read())
        if not isinstance(search_space, Mapping):
            raise ValueError(f"search_space must be of type str or DictConfig. Got {type(search_space)}.")
        hps = search_space.get("hyperparameters", [])
        if isinstance(hps, Mapping):  # hydra form: {name: {key: value}}
            hps = [{**dict(cfg), "name": name} for name, cfg in hps.items()]
        cs = ConfigurationSpace(name=search_space.get("name"))
        cs.add_hyperparameters([_feature_from_json(dict(h)) for h in hps])
    if seed is not None:
        cs.seed(seed=seed)
    return cs
