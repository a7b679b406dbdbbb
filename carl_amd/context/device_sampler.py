"""Dense context sets produced on the device.

``ContextSampler.sample_contexts`` (carl/context/sampler.py:45-61) draws every sampled feature
per context and fills the rest with defaults as Python dicts; at 65 k - 131 k contexts that is
seconds of interpreter time before the first step.  ``sample_context_table_device`` produces the
same kind of context set -- every feature of the env's context space, sampled ones from their
distributions, the others at their defaults -- directly as the ``[F][C]`` float32 table in HBM
that the step kernels read (``carl_sample_contexts``, include/carl_amd.h), and
``verify_table_device`` is ``ContextSpace.verify_context`` (context_space.py:54-59) for it.

The device stream is Philox keyed by (seed, global context id, feature): reproducible, independent
of how contexts are sharded over GPUs, and NOT the reference's NumPy stream -- use
``ContextSampler.sample_context_table`` (host) when the reference's exact draws matter.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Sequence

import numpy as np

from carl_amd import _lib
from carl_amd.context.context_space import ContextSpace
from carl_amd.context.features import (
    CategoricalContextFeature,
    ContextFeature,
    NormalFloatContextFeature,
    UniformFloatContextFeature,
    UniformIntegerContextFeature,
)
from carl_amd.context.table import ContextTable

_F32_MAX = 3.4028234663852886e38


def _bound(v: float) -> float:
    v = float(v)
    if math.isinf(v):
        return v
    return max(-_F32_MAX, min(_F32_MAX, v))


def feature_spec(feature: ContextFeature, sampled: bool) -> _lib.FeatureSpec:
    """One ``carl_feature_spec_t``: the feature's distribution if ``sampled``, else its default as a
    constant; bounds are always carried (verification)."""
    sp = _lib.FeatureSpec()
    if isinstance(feature, CategoricalContextFeature):
        try:
            choices = [float(c) for c in feature.choices]
        except (TypeError, ValueError):
            raise ValueError(f"{feature.name}: the device table holds numeric categorical choices only")
        if len(choices) > _lib.MAX_CHOICES:
            raise ValueError(f"{feature.name}: more than {_lib.MAX_CHOICES} choices")
        if sampled and not np.allclose(feature.probabilities, 1.0 / len(choices)):
            raise ValueError(f"{feature.name}: weighted categorical features are sampled on the host only")
        sp.n_choices = len(choices)
        for k, c in enumerate(choices):
            sp.choices[k] = c
        sp.lower, sp.upper = min(choices), max(choices)
        sp.value = float(feature.default_value)
        # a constant categorical keeps kind CATEGORICAL only for verification: mark via n_choices
        sp.kind = _lib.FEAT_CATEGORICAL if sampled else _lib.FEAT_CONSTANT
        return sp
    sp.lower, sp.upper = _bound(feature.lower), _bound(feature.upper)
    sp.value = float(feature.default_value)
    if not sampled:
        sp.kind = _lib.FEAT_CONSTANT
    elif isinstance(feature, NormalFloatContextFeature):
        sp.kind, sp.mu, sp.sigma = _lib.FEAT_NORMAL_FLOAT, feature.mu, feature.sigma
    elif isinstance(feature, UniformIntegerContextFeature):
        if feature.log:  # carl_feature_spec_t has a log scale for uniform floats only
            raise ValueError(f"{feature.name}: log-scale integer features are sampled on the host only")
        sp.kind = _lib.FEAT_UNIFORM_INT
    elif isinstance(feature, UniformFloatContextFeature):
        sp.kind, sp.log_scale = _lib.FEAT_UNIFORM_FLOAT, int(bool(feature.log))
    else:
        raise ValueError(f"{feature.name}: unsupported distribution {type(feature).__name__}")
    return sp


def build_specs(context_space: ContextSpace, distributions: Sequence[ContextFeature]):
    """Specs in the context space's feature order (= table row order); ``distributions`` override
    the space's features of the same name, exactly like ``default_context | sampled``
    (sampler.py:57-61)."""
    names = list(context_space.context_feature_names)
    dist = {d.name: d for d in distributions}
    unknown = [n for n in dist if n not in names]
    if unknown:
        raise ValueError(f"Unknown context features {unknown}; known: {names}")
    specs = (_lib.FeatureSpec * len(names))()
    for j, n in enumerate(names):
        specs[j] = feature_spec(dist[n], True) if n in dist else feature_spec(context_space.context_space[n], False)
    return names, specs


class DeviceContextTable(ContextTable):
    """A context set living in HBM as the engine's ``[F][C]`` float32 table.  The ``Contexts``
    mapping view / ``values_2d`` download the table on first use."""

    def __init__(self, names: Sequence[str], tensor, keys=None):
        if tensor.ndim != 2 or tensor.shape[0] != len(names):
            raise ValueError(f"tensor must be [{len(names)}, C], got {tuple(tensor.shape)}")
        self.names = list(names)
        self.tensor = tensor
        self._host = None
        self._keys = list(range(tensor.shape[1])) if keys is None else list(keys)
        self._pos = None

    @property
    def values_2d(self) -> np.ndarray:  # [C, F] float64, like ContextTable
        if self._host is None:
            self._host = self.tensor.t().to("cpu").double().numpy()
        return self._host

    def __len__(self) -> int:
        return int(self.tensor.shape[1])


def _upload_specs(specs, device):
    import torch

    raw = np.frombuffer(bytes(specs), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device)


def sample_context_table_device(context_space: ContextSpace, distributions: Sequence[ContextFeature], n_contexts: int,
                                seed: int, device="cuda", context_offset: int = 0) -> DeviceContextTable:
    import torch

    lib = _lib.load()
    names, specs = build_specs(context_space, distributions)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.CarlHipError("sample_context_table_device needs an MI355X device (no CPU path)")
    specs_dev = _upload_specs(specs, dev)
    table = torch.empty((len(names), int(n_contexts)), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.carl_sample_contexts(specs_dev.data_ptr(), specs, len(names), int(n_contexts), int(n_contexts),
                                            int(context_offset), int(seed) & (2**64 - 1), table.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream))
    return DeviceContextTable(names, table, keys=range(int(context_offset), int(context_offset) + int(n_contexts)))


def verify_table_device(context_space: ContextSpace, table: DeviceContextTable) -> int:
    """Number of entries outside their feature's bounds / choices (0 = the table verifies)."""
    import torch

    lib = _lib.load()
    names = list(context_space.context_feature_names)
    if list(table.names) != names:
        raise ValueError("table feature order differs from the context space")
    specs = (_lib.FeatureSpec * len(names))()
    for j, n in enumerate(names):
        f = context_space.context_space[n]
        specs[j] = feature_spec(f, isinstance(f, CategoricalContextFeature))  # categorical: check membership
    t = table.tensor
    specs_dev = _upload_specs(specs, t.device)
    n_bad = torch.zeros(1, dtype=torch.int32, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(lib.carl_verify_contexts(specs_dev.data_ptr(), specs, len(names), int(t.shape[1]), int(t.stride(0)),
                                            t.data_ptr(), n_bad.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return int(n_bad.item())
