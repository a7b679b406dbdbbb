"""Runs the REFERENCE's own unit tests for the pure-host context modules against this package, with
``carl`` aliased to ``carl_amd`` (and the two third-party names those tests import -- ``gymnasium.spaces``,
``ConfigSpace.ConfigurationSpace`` / ``omegaconf.DictConfig`` -- aliased to the stand-ins the product itself
uses: ``carl_amd.spaces``, ``carl_amd.context.features.ConfigurationSpace``, ``dict``).  The files are executed
from /root/reference where they lie (never copied); skipped where the reference tree is absent (GPU box).

  test/test_context_sampler.py        2 tests   ContextSampler draws / defaults
  test/test_context_bounds.py         1 test    get_context_bounds
  test/test_context_space.py          ...       ContextSpace (verify, defaults, bounds, gymnasium-space shapes)
  test/test_search_space_encoding.py  ...       search_space_to_config_space

The env-constructing reference tests (test_CARLEnv.py, test_context_selector.py, test_gymnasium_envs.py,
test_brax_env.py, test_language_goals.py) need a device; their assertions are mirrored in
tests/test_gpu_env_api.py and tests/test_gpu_brax.py."""
import importlib
import importlib.util
import os
import sys
import types
import unittest

import pytest

REF = "/root/reference/test"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _alias_modules():
    import carl_amd  # noqa: F401
    import carl_amd.context.context_space  # noqa: F401
    import carl_amd.context.sampler  # noqa: F401
    import carl_amd.context.search_space_encoding  # noqa: F401
    import carl_amd.context.selection  # noqa: F401
    import carl_amd.context.utils  # noqa: F401
    import carl_amd.spaces as spaces
    import carl_amd.utils.types  # noqa: F401
    from carl_amd.context.features import ConfigurationSpace

    added = {}
    for name, mod in list(sys.modules.items()):
        if name == "carl_amd" or name.startswith("carl_amd."):
            added["carl" + name[len("carl_amd"):]] = mod
    gym = types.ModuleType("gymnasium")
    gym.spaces = spaces
    added["gymnasium"] = gym
    added["gymnasium.spaces"] = spaces
    cs = types.ModuleType("ConfigSpace")
    cs.ConfigurationSpace = ConfigurationSpace
    added["ConfigSpace"] = cs
    oc = types.ModuleType("omegaconf")
    oc.DictConfig = dict
    added["omegaconf"] = oc
    return added


@pytest.mark.parametrize("name", ["test_context_sampler", "test_context_bounds", "test_context_space",
                                  "test_search_space_encoding"])
def test_reference_test_file_passes_on_the_mirror(name, monkeypatch):
    for k, v in _alias_modules().items():
        if k not in sys.modules:
            monkeypatch.setitem(sys.modules, k, v)
    spec = importlib.util.spec_from_file_location("_ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
    assert suite.countTestCases() > 0
    result = unittest.TextTestRunner(verbosity=0, stream=open(os.devnull, "w")).run(suite)
    problems = [f"{t}: {tb.splitlines()[-1]}" for t, tb in result.failures + result.errors]
    assert not problems, problems
