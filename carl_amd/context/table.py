"""Dense context sets.

The reference holds a context set as a dict of dicts (carl/utils/types.py:5-6) and
fills defaults per context in Python (carl/envs/carl_env.py:135-137).  At 65 536+
contexts that is seconds of interpreter time before the first step (SURVEY.md
section 3.5), so the engine's native form is a dense ``[C, F]`` float64 matrix in
feature-table order, with a read-only ``Mapping`` view that yields the reference's
``{key: {feature: value}}`` shape on demand.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any, Sequence

import numpy as np


class ContextTable(Mapping):
    """``Contexts``-compatible view over a dense ``[C, F]`` matrix."""

    def __init__(self, names: Sequence[str], values: np.ndarray, keys: Sequence[Any] | None = None):
        values = np.asarray(values, dtype=np.float64)
        if values.ndim != 2 or values.shape[1] != len(names):
            raise ValueError(f"values must be [C, {len(names)}], got {values.shape}")
        self.names = list(names)
        self.values_2d = values
        self._keys = list(range(values.shape[0])) if keys is None else list(keys)
        if len(self._keys) != values.shape[0]:
            raise ValueError("one key per context row is required")
        self._pos = None

    # -- Mapping protocol ---------------------------------------------------
    def __len__(self) -> int:
        return self.values_2d.shape[0]

    def __iter__(self):
        return iter(self._keys)

    def __getitem__(self, key):
        if self._pos is None:
            self._pos = {k: i for i, k in enumerate(self._keys)}
        row = self.values_2d[self._pos[key]]
        return {n: float(v) for n, v in zip(self.names, row)}

    def keys(self):
        return list(self._keys)

    # -- dense access ---------------------------------------------------------
    def column(self, name: str) -> np.ndarray:
        return self.values_2d[:, self.names.index(name)]

    def reordered(self, names: Sequence[str], defaults: Mapping[str, float]) -> "ContextTable":
        """Same contexts in another feature order, missing columns filled with defaults."""
        out = np.empty((len(self), len(names)), dtype=np.float64)
        for j, n in enumerate(names):
            out[:, j] = self.values_2d[:, self.names.index(n)] if n in self.names else float(defaults[n])
        return ContextTable(names, out, self._keys)

    @staticmethod
    def from_contexts(contexts: Mapping, names: Sequence[str], defaults: Mapping[str, Any]) -> "ContextTable":
        """Dict-of-dicts -> dense, filling defaults (carl_env.py:135-137 semantics)."""
        if isinstance(contexts, ContextTable):
            extra = [n for n in contexts.names if n not in names]
            if extra:
                raise ValueError(f"Unknown context features {extra}")
            return contexts.reordered(names, defaults)
        keys = list(contexts.keys())
        out = np.empty((len(keys), len(names)), dtype=np.float64)
        col = {n: j for j, n in enumerate(names)}
        base = np.array([float(defaults[n]) for n in names], dtype=np.float64)
        for i, k in enumerate(keys):
            out[i] = base
            for n, v in contexts[k].items():
                if n not in col:
                    raise ValueError(f"Unknown context feature {n!r}; known: {list(names)}")
                out[i, col[n]] = float(v)
        return ContextTable(names, out, keys)
