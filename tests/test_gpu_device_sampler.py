"""carl_sample_contexts / carl_verify_contexts (through the C ABI) against the CPU restatement,
and an env built on a device-resident context table against the same env built from the
downloaded host copy."""
import numpy as np
import pytest
import torch

from carl_amd.context.context_space import NormalFloatContextFeature, UniformFloatContextFeature
from carl_amd.context.device_sampler import build_specs, sample_context_table_device, verify_table_device
from carl_amd.context.sampler import ContextSampler
from carl_amd.context.selection import StaticSelector
from carl_amd.context.table import ContextTable
from oracle import oracle as O
from test_device_sampler import DISTS, SPACE

pytestmark = pytest.mark.gpu


def test_device_table_matches_restatement(device):
    names, specs = build_specs(SPACE, DISTS)
    n = 100_003  # ragged last workgroup
    dt = sample_context_table_device(SPACE, DISTS, n, seed=99, device=device, context_offset=5_000_000_000)
    got = dt.tensor.cpu().numpy()
    want = O.sample_contexts(specs, n, seed=99, context_offset=5_000_000_000)
    assert got.shape == want.shape == (6, n) and dt.names == names
    for j, name in enumerate(names):
        if name in ("length", "mass"):  # float32 log/sqrt/sincos/exp on the device vs double on the host
            np.testing.assert_allclose(got[j], want[j], rtol=3e-6, atol=3e-7, err_msg=name)
        else:                           # same float32 expression: bit-exact
            np.testing.assert_array_equal(got[j], want[j], err_msg=name)
    assert verify_table_device(SPACE, dt) == 0
    dt.tensor[0, 17] = -1.0  # below the space's lower bound (0.1); the upper bound is +inf
    dt.tensor[4, 3] = 7.0
    dt.tensor[3, 1] = float("nan")
    assert verify_table_device(SPACE, dt) == 3


def test_shard_is_a_slice_of_the_global_set(device):
    full = sample_context_table_device(SPACE, DISTS, 8192, seed=5, device=device).tensor
    part = sample_context_table_device(SPACE, DISTS, 2048, seed=5, device=device, context_offset=4096).tensor
    assert torch.equal(full[:, 4096:6144], part)


def test_env_on_device_table_equals_env_on_host_copy(device):
    from carl_amd.envs import CARLBraxAnt, CARLPendulum

    n = 4096
    sampler = ContextSampler([UniformFloatContextFeature("g", 1, 20), NormalFloatContextFeature("l", 1.0, 0.3, 0.5, 2.0)],
                             CARLPendulum.get_context_space(), seed=3)
    dt = sampler.sample_context_table_device(n, device)
    assert len(dt) == n and dt.tensor.shape == (len(CARLPendulum.get_context_features()), n)
    host = ContextTable(dt.names, dt.values_2d)
    envs = [CARLPendulum(contexts=c, num_envs=n, device=device, context_selector=StaticSelector, seed=1)
            for c in (dt, host)]
    obs = [e.reset(seed=1)[0] for e in envs]
    assert torch.equal(obs[0]["obs"], obs[1]["obs"]) and torch.equal(obs[0]["context"]["g"], obs[1]["context"]["g"])
    assert envs[0].env.ctx_table.data_ptr() == dt.tensor.data_ptr()  # adopted, not copied
    a = torch.rand(n, device=device) * 4 - 2
    for _ in range(5):
        out = [e.step(a) for e in envs]
        assert torch.equal(out[0][0]["obs"], out[1][0]["obs"]) and torch.equal(out[0][1], out[1][1])
    # Brax: categorical target_direction sampled on the device feeds the goal-free env unchanged
    sampler = ContextSampler([UniformFloatContextFeature("gravity", -15, -5), UniformFloatContextFeature("mass_torso", 5, 15)],
                             CARLBraxAnt.get_context_space(), seed=4)
    ant = CARLBraxAnt(contexts=sampler.sample_context_table_device(256, device), batch_size=256, device=device,
                      context_selector=StaticSelector)
    o, _ = ant.reset(seed=0)
    assert o["obs"].shape == (256, 27) and float(o["context"]["gravity"].max()) <= -5.0
    o, r, te, tr, _ = ant.step(torch.zeros((256, 8), device=device))
    assert torch.isfinite(o["obs"]).all()
