"""Minimal observation/action spaces.

The reference describes observations with ``gymnasium.spaces`` (carl_env.py:182-187,
context_space.py:166-188).  gymnasium is not a dependency of this build, so the three
space kinds CARL uses are provided with the same attribute surface (``shape``,
``dtype``, ``low``/``high``, ``n``, ``spaces``, ``contains``, ``sample``); when
gymnasium IS importable its own classes are used instead so existing wrappers
(``FlattenObservation`` ...) accept the spaces unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Mapping

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box, Dict, Discrete, Space  # type: ignore

    HAVE_GYMNASIUM = True
except Exception:  # ModuleNotFoundError
    HAVE_GYMNASIUM = False

    class Space:
        shape: tuple | None = None
        dtype: Any = None

        def __init__(self, seed: int | None = None):
            self._rng = np.random.default_rng(seed)

        def seed(self, seed: int | None = None):
            self._rng = np.random.default_rng(seed)

        def contains(self, x) -> bool:
            raise NotImplementedError

        def __contains__(self, x) -> bool:
            return self.contains(x)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            super().__init__(seed)
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
                if shape == ():
                    shape = (1,)
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape).astype(self.dtype)
            self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return bool(
                x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)
            )

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e6)
            hi = np.where(np.isfinite(self.high), self.high, 1e6)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        def __eq__(self, other):
            return (isinstance(other, Box) and self.shape == other.shape
                    and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    class Discrete(Space):
        def __init__(self, n: int, seed=None, start: int = 0):
            super().__init__(seed)
            self.n = int(n)
            self.start = int(start)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def contains(self, x) -> bool:
            try:
                v = int(x)
            except Exception:
                return False
            return v == x and self.start <= v < self.start + self.n

        def sample(self):
            return int(self._rng.integers(self.start, self.start + self.n))

        def __repr__(self):
            return f"Discrete({self.n})"

        def __eq__(self, other):
            return isinstance(other, Discrete) and self.n == other.n and self.start == other.start

    class Dict(Space):
        def __init__(self, spaces: Mapping[str, Space] | None = None, seed=None, **kw):
            super().__init__(seed)
            self.spaces = OrderedDict(spaces or {})
            self.spaces.update(kw)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def __len__(self):
            return len(self.spaces)

        def contains(self, x) -> bool:
            return (isinstance(x, Mapping) and x.keys() == self.spaces.keys()
                    and all(self.spaces[k].contains(x[k]) for k in x))

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {s}" for k, s in self.spaces.items()) + ")"


def batch_space(space: "Space", n: int) -> "Space":
    """``gymnasium.vector.utils.batch_space`` for the kinds used here (the reference's
    only batched API does this: carl/envs/brax/wrappers.py:111-118)."""
    if isinstance(space, Box):
        return Box(low=np.repeat(space.low[None], n, axis=0), high=np.repeat(space.high[None], n, axis=0),
                   dtype=space.dtype)
    if isinstance(space, Discrete):
        return Box(low=space.start, high=space.start + space.n - 1, shape=(n,), dtype=np.int64)
    if isinstance(space, Dict):
        return Dict({k: batch_space(s, n) for k, s in space.spaces.items()})
    raise TypeError(f"cannot batch {type(space)}")
