"""Context selectors (API of the reference's carl/context/selection.py:11-180).

Two uses: (1) host-side objects with the reference's attributes (``contexts``,
``context_ids``, ``contexts_keys``, ``n_calls``, ``context_id``, ``context_key``,
``select()``) -- exact for one env; (2) a declaration of the per-lane rule the
device applies when lanes auto-reset (``device_rule``): the HIP reset path advances
each lane's context id itself, so round robin / static / random need no host
round trip.  Custom selectors run on the host (no in-kernel auto-reset).
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Tuple

import numpy as np

from carl_amd.utils.types import Context, Contexts

# device-side selector rules (values match include/carl_amd.h CARL_SEL_*)
SEL_STATIC, SEL_ROUND_ROBIN, SEL_RANDOM, SEL_HOST = 0, 1, 2, 3


class AbstractSelector:
    """A selector walks over a context set.  Public state, as the reference exposes it: ``contexts`` (the
    set), ``contexts_keys`` (its keys in order), ``context_ids`` (positions 0..n-1), ``context_id``
    (position of the current context, ``None`` before the first ``select``), ``n_calls``.  Subclasses say
    which position comes next (``_next_id``); ``device_rule`` names the same rule for the in-kernel
    auto-reset."""

    device_rule: int = SEL_HOST

    def __init__(self, contexts: Contexts):
        self.contexts = contexts
        self.contexts_keys: List[Any] = [*contexts.keys()]
        self.context_ids: List[int] = list(np.arange(len(self.contexts_keys)))
        self.context_id: Optional[int] = None
        self.n_calls = 0

    # -- what subclasses provide ---------------------------------------------------------
    def _next_id(self) -> int:
        raise NotImplementedError

    def _select(self) -> Tuple[Context, int]:
        """(context, position) of the next context; also moves ``context_id`` there"""
        self.context_id = self._next_id()
        return self._at(self.context_id), self.context_id

    # -- API -----------------------------------------------------------------------------
    def _at(self, position: int) -> Context:
        return self.contexts[self.contexts_keys[position]]

    def select(self) -> Context:
        context, self.context_id = self._select()
        self.n_calls += 1
        return context

    @property
    def context_key(self) -> Any | None:
        """Key of the current context.  Position 0 reports ``None`` as well as "nothing selected yet":
        the reference tests the id for truth (selection.py:91, Quirk S2), and callers may rely on it."""
        return self.contexts_keys[self.context_id] if self.context_id else None


class RandomSelector(AbstractSelector):
    """A uniformly random position at every reset.  On the host this is the global, unseeded
    ``np.random`` stream the reference uses (selection.py:105); on the device, the lane's Philox stream."""

    device_rule = SEL_RANDOM

    def _next_id(self) -> int:
        return np.random.choice(self.context_ids)


class RoundRobinSelector(AbstractSelector):
    """Positions 0, stride, 2 stride, ... modulo the set size (selection.py:116-122 is stride 1)."""

    device_rule = SEL_ROUND_ROBIN

    def __init__(self, contexts: Contexts, stride: int = 1):
        super().__init__(contexts)
        self.stride = stride

    def _next_id(self) -> int:
        previous = -self.stride if self.context_id is None else self.context_id
        return (previous + self.stride) % len(self.contexts)


class StaticSelector(AbstractSelector):
    """Stays where it is; starts at the first context (selection.py:131-136)."""

    device_rule = SEL_STATIC

    def _next_id(self) -> int:
        return self.context_ids[0] if self.context_id is None else self.context_id


class CustomSelector(AbstractSelector):
    """``selector_function(selector) -> (context, position)`` decides (selection.py:139-180).  Runs on the
    host only: a batched env with this selector cannot auto-reset inside the kernel."""

    device_rule = SEL_HOST

    def __init__(self, contexts: Contexts, selector_function: Callable[[AbstractSelector], Tuple[Context, int]]):
        super().__init__(contexts)
        self.selector_function = selector_function

    def _select(self) -> Tuple[Context, int]:
        context, self.context_id = self.selector_function(self)
        return context, self.context_id
