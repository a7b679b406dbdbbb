"""Bounds helper (reference: carl/context/utils.py:6-35)."""
from typing import Any, Dict, List, Tuple, Type

import numpy as np


def get_context_bounds(
    context_keys: List[str], context_bounds: Dict[str, Tuple[float, float, Type[Any]]]
) -> Tuple[np.ndarray, np.ndarray]:
    """Lower / upper bound arrays for ``context_keys`` from ``{name: (lo, hi, type)}``."""
    lower_bounds = np.array([context_bounds[k][0] for k in context_keys], dtype=np.float64)
    upper_bounds = np.array([context_bounds[k][1] for k in context_keys], dtype=np.float64)
    return lower_bounds, upper_bounds
