"""Seeded context sampler (API of the reference's carl/context/sampler.py:11-61).

``sample_contexts(n)`` returns the reference's ``{i: {feature: value}}`` dict;
``sample_context_table(n)`` returns the same draws as a dense ``ContextTable``
without building n Python dicts -- the form the engine uploads.
"""
from __future__ import annotations

from collections.abc import Mapping

import numpy as np

from carl_amd.context.context_space import ContextFeature, ContextSpace
from carl_amd.context.features import ConfigurationSpace
from carl_amd.context.search_space_encoding import search_space_to_config_space
from carl_amd.context.table import ContextTable
from carl_amd.utils.types import Context, Contexts


class ContextSampler(ConfigurationSpace):
    """A configuration space over the features that VARY; every other feature of ``context_space`` stays at
    its default in the sampled contexts.  ``context_distributions``: a list or a dict of context features,
    or a search-space description (a string, or an omegaconf-style mapping with a "hyperparameters" entry)
    that ``search_space_to_config_space`` understands."""

    def __init__(self, context_distributions, context_space: ContextSpace, seed: int, name: str | None = None):
        super().__init__(name=name, seed=seed)
        self.context_distributions = context_distributions
        self.context_space = context_space
        self._device_seed = int(seed) if seed is not None else 0  # key of the device-side Philox stream
        self.add_context_features(self._varying_features(context_distributions))
        self.context_feature_names = [feature.name for feature in self.get_context_features()]

    @staticmethod
    def _varying_features(spec) -> list[ContextFeature]:
        if isinstance(spec, dict):  # name -> feature (a plain dict is never read as a search space)
            return list(spec.values())
        if isinstance(spec, list):
            return spec
        if isinstance(spec, str) or (isinstance(spec, Mapping) and "hyperparameters" in spec):
            return search_space_to_config_space(spec).get_hyperparameters()
        raise ValueError(f"Unknown type `{type(spec)}` for `context_distributions`.")

    def add_context_features(self, context_features) -> None:
        self.add_hyperparameters(context_features)

    def get_context_features(self) -> list[ContextFeature]:
        return list(self.values())

    def sample_contexts(self, n_contexts: int) -> Contexts:
        """``{0: context, 1: context, ...}`` with ``n_contexts`` entries"""
        return dict(enumerate(self._sample_contexts(size=n_contexts)))

    def _sample_contexts(self, size: int = 1) -> list[Context]:
        drawn = self.sample_configuration(size=size)
        if size == 1:  # a single configuration comes back bare
            drawn = [drawn]
        defaults = self.context_space.get_default_context()
        # defaults first, then the drawn values (a drawn name the space does not know is appended)
        return [{**defaults, **dict(configuration)} for configuration in drawn]

    def sample_context_table(self, n_contexts: int) -> ContextTable:
        """Same RNG consumption and values as ``sample_contexts`` (numeric features only)."""
        cols = self.sample_vectors(n_contexts)
        names = list(self.context_space.context_feature_names)
        extra = [k for k in cols if k not in names]
        names = names + extra  # reference: default_context | dict(C) appends unknown keys
        base = self.context_space.get_default_context()
        out = np.empty((n_contexts, len(names)), dtype=np.float64)
        for j, n in enumerate(names):
            out[:, j] = np.asarray(cols[n], dtype=np.float64) if n in cols else float(base[n])
        return ContextTable(names, out)

    def sample_context_table_device(self, n_contexts: int, device="cuda", context_offset: int = 0):
        """The same kind of context set, produced directly in HBM by ``carl_sample_contexts``
        (Philox stream keyed by this sampler's seed; see carl_amd/context/device_sampler.py)."""
        from carl_amd.context.device_sampler import sample_context_table_device

        return sample_context_table_device(self.context_space, self.get_context_features(), n_contexts,
                                           self._device_seed, device, context_offset)
