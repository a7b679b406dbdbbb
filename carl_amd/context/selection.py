"""Context selectors (API of the reference's carl/context/selection.py:11-180).

Two uses: (1) host-side objects with the reference's attributes (``contexts``,
``context_ids``, ``contexts_keys``, ``n_calls``, ``context_id``, ``context_key``,
``select()``) -- exact for one env; (2) a declaration of the per-lane rule the
device applies when lanes auto-reset (``device_rule``): the HIP reset path advances
each lane's context id itself, so round robin / static / random need no host
round trip.  Custom selectors run on the host (no in-kernel auto-reset).
"""
from __future__ import annotations

from abc import abstractmethod
from typing import Any, Callable, List, Optional, Tuple

import numpy as np

from carl_amd.utils.types import Context, Contexts

# device-side selector rules (values match include/carl_amd.h CARL_SEL_*)
SEL_STATIC, SEL_ROUND_ROBIN, SEL_RANDOM, SEL_HOST = 0, 1, 2, 3


class AbstractSelector(object):
    """Base class; the context is chosen in ``select``, not in ``__init__``."""

    device_rule: int = SEL_HOST

    def __init__(self, contexts: Contexts):
        self.contexts: Contexts = contexts
        self.context_ids: List[int] = list(np.arange(len(contexts)))
        self.contexts_keys: List[Any] = list(contexts.keys())
        self.n_calls: int = 0
        self.context_id: Optional[int] = None

    @abstractmethod
    def _select(self) -> Tuple[Context, int]:
        ...

    def select(self) -> Context:
        context, context_id = self._select()
        self.context_id = context_id
        self.n_calls += 1
        return context

    @property
    def context_key(self) -> Any | None:
        # Quirk S2 of the reference (selection.py:91): id 0 is falsy -> None
        if self.context_id:
            return self.contexts_keys[self.context_id]
        return None


class RandomSelector(AbstractSelector):
    """Uniformly random context each reset.  The reference draws from the global,
    unseeded ``np.random`` (selection.py:105); so does the host object.  On device
    the draw is the lane's Philox stream (reproducible from the env seed)."""

    device_rule = SEL_RANDOM

    def _select(self) -> Tuple[Context, int]:
        context_id = np.random.choice(self.context_ids)
        return self.contexts[self.contexts_keys[context_id]], context_id


class RoundRobinSelector(AbstractSelector):
    """Next context in order, wrapping (selection.py:116-122)."""

    device_rule = SEL_ROUND_ROBIN

    def __init__(self, contexts: Contexts, stride: int = 1):
        super().__init__(contexts)
        self.stride = stride

    def _select(self) -> Tuple[Context, int]:
        if self.context_id is None:
            self.context_id = -self.stride
        self.context_id = (self.context_id + self.stride) % len(self.contexts)
        return self.contexts[self.contexts_keys[self.context_id]], self.context_id


class StaticSelector(AbstractSelector):
    """Never changes the context (selection.py:131-136)."""

    device_rule = SEL_STATIC

    def _select(self) -> Tuple[Context, int]:
        if self.context_id is None:
            self.context_id = self.context_ids[0]
        return self.contexts[self.contexts_keys[self.context_id]], self.context_id


class CustomSelector(AbstractSelector):
    """User function ``f(selector) -> (context, context_id)`` (selection.py:139-180)."""

    device_rule = SEL_HOST

    def __init__(self, contexts: Contexts,
                 selector_function: Callable[[AbstractSelector], Tuple[Context, int]]):
        super().__init__(contexts=contexts)
        self.selector_function = selector_function

    def _select(self) -> Tuple[Context, int]:
        context, context_id = self.selector_function(self)
        self.context_id = context_id
        return context, context_id
