"""Context space: defaults, bounds, verification, sampling.

API of the reference's carl/context/context_space.py:31-229, on this build's own
feature types (ConfigSpace is not used).  Gymnasium spaces come from
``carl_amd.spaces`` (gymnasium itself is used when importable).
"""
from __future__ import annotations

from typing import List

import numpy as np

from carl_amd import spaces
from carl_amd.context.features import (
    CategoricalContextFeature,
    ContextFeature,
    NormalFloatContextFeature,
    NumericalContextFeature,
    UniformFloatContextFeature,
    UniformIntegerContextFeature,
)
from carl_amd.context.table import ContextTable
from carl_amd.utils.types import Context, Contexts

__all__ = [
    "ContextFeature", "NumericalContextFeature", "NormalFloatContextFeature",
    "UniformFloatContextFeature", "UniformIntegerContextFeature",
    "CategoricalContextFeature", "ContextSpace",
]


class ContextSpace(object):
    def __init__(self, context_space: dict[str, ContextFeature]) -> None:
        self.context_space = context_space

    @property
    def context_feature_names(self) -> list[str]:
        return list(self.context_space.keys())

    def insert_defaults(self, context: Context, context_keys: List[str] | None = None) -> Context:
        """Defaults for every feature (or only ``context_keys``), overridden by
        ``context`` (reference :54-80, including Quirk S1: keys of ``context`` that
        are outside ``context_keys`` still come through)."""
        filled = self.get_default_context()
        if context_keys:
            filled = {key: filled[key] for key in context_keys}
        filled.update(context)
        return filled

    def verify_context(self, context: Context) -> bool:
        """Names known and numerical values within bounds (reference :82-113)."""
        for name, value in context.items():
            if name not in self.context_space:
                return False
            cf = self.context_space[name]
            if isinstance(cf, NumericalContextFeature) and not (cf.lower <= value <= cf.upper):
                return False
        return True

    def get_default_context(self) -> Context:
        return {cf.name: cf.default_value for cf in self.context_space.values()}

    def get_lower_and_upper_bound(self, context_feature_name: str) -> tuple[float, float]:
        cf = self.context_space[context_feature_name]
        return (cf.lower, cf.upper)

    def to_gymnasium_space(self, context_feature_names: List[str] | None = None,
                           as_dict: bool = False) -> spaces.Space:
        """Dict of Box/Discrete (``as_dict``) or one float32 Box (reference :145-188)."""
        if context_feature_names is None:
            context_feature_names = self.context_feature_names
        if as_dict:
            sub = {}
            for name in context_feature_names:
                cf = self.context_space[name]
                if isinstance(cf, NumericalContextFeature):
                    sub[cf.name] = spaces.Box(low=cf.lower, high=cf.upper)
                else:
                    sub[cf.name] = spaces.Discrete(len(cf.choices))
            return spaces.Dict(sub)
        low = np.array([self.context_space[n].lower for n in context_feature_names])
        high = np.array([self.context_space[n].upper for n in context_feature_names])
        return spaces.Box(low=low, high=high, dtype=np.float32)

    def sample_contexts(self, context_keys: List[str] | None = None, size: int = 1) -> Context | List[Contexts]:
        """``size`` contexts with EVERY feature drawn by ``rvs()`` (reference :190-229;
        ``context_keys`` is only validated -- Quirk S1)."""
        if context_keys is None:
            context_keys = self.context_space.keys()
        else:
            for key in context_keys:
                if key not in self.context_space.keys():
                    raise ValueError(f"Invalid context feature name: {key}")
        contexts = []
        for _ in range(size):
            context = {cf.name: cf.rvs() for cf in self.context_space.values()}
            contexts.append(self.insert_defaults(context, context_keys))
        return contexts[0] if size == 1 else contexts

    # ---- dense helpers (no reference counterpart) ---------------------------
    def default_row(self) -> np.ndarray:
        return np.array([float(cf.default_value) for cf in self.context_space.values()], dtype=np.float64)

    def to_table(self, contexts) -> ContextTable:
        """Any ``Contexts`` mapping -> dense table in this space's feature order."""
        return ContextTable.from_contexts(contexts, self.context_feature_names, self.get_default_context())

    def verify_table(self, table: ContextTable) -> np.ndarray:
        """Vectorised ``verify_context``: bool per context row."""
        ok = np.ones(len(table), dtype=bool)
        for j, name in enumerate(table.names):
            if name not in self.context_space:
                ok[:] = False
                break
            cf = self.context_space[name]
            if isinstance(cf, NumericalContextFeature):
                col = table.values_2d[:, j]
                ok &= (col >= cf.lower) & (col <= cf.upper)
        return ok
