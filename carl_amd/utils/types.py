"""Names the reference API annotates with (carl/utils/types.py:5-8): a context is a mapping from feature
name to value, a context set maps ids to contexts, a vector is a list or an array."""
from __future__ import annotations

import typing as t

import numpy as np

ObsType = t.TypeVar("ObsType")
Context = t.Dict[str, t.Any]          # feature name -> value
Contexts = t.Dict[t.Any, Context]     # context id -> context
Vector = t.Union[t.List[t.Any], np.ndarray]
