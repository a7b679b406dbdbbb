"""Host selector objects against ids produced by RUNNING the reference's selectors
(tests/golden/selector_sequences.json <- carl/context/selection.py).  CPU-only; the
env-level selector tests of the reference (test/test_context_selector.py) need an engine
and live in test_gpu_env_api.py."""
import json
import os

import numpy as np
import pytest

from carl_amd.context.selection import (
    AbstractSelector,
    CustomSelector,
    RandomSelector,
    RoundRobinSelector,
    StaticSelector,
)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "selector_sequences.json")))


def run(selector, n):
    ids, calls, keys = [], [], []
    for _ in range(n):
        ctx = selector.select()
        assert ctx is selector.contexts[selector.contexts_keys[selector.context_id]]
        ids.append(int(selector.context_id))
        calls.append(selector.n_calls)
        keys.append(selector.context_key)
    return {"context_id": ids, "n_calls": calls, "context_key": keys}


@pytest.mark.parametrize("n_ctx", [1, 2, 3, 5, 7])
def test_round_robin_and_static_match_reference(gold, n_ctx):
    contexts = {chr(ord("a") + i): {"x": float(i)} for i in range(n_ctx)}
    assert run(RoundRobinSelector(contexts=contexts), 12) == gold[f"round_robin_{n_ctx}"]
    assert run(StaticSelector(contexts=contexts), 6) == gold[f"static_{n_ctx}"]


def test_custom_selector_matches_reference(gold):
    def fn(inst):
        cid = 1 if inst.n_calls == 0 else 0
        return inst.contexts[inst.contexts_keys[cid]], cid

    contexts = {k: {"x": 0.0} for k in "abc"}
    assert run(CustomSelector(contexts=contexts, selector_function=fn), 5) == gold["custom_first1_then0"]


def test_context_key_quirk_and_initial_state():
    s = RoundRobinSelector(contexts={"a": {}, "b": {}})
    assert s.context_id is None and s.context_key is None and s.n_calls == 0
    assert s.context_ids == [0, 1] and s.contexts_keys == ["a", "b"]
    s.select()
    assert s.context_id == 0 and s.context_key is None  # Quirk S2: id 0 is falsy
    s.select()
    assert s.context_key == "b"


def test_random_selector_uses_global_numpy_state():
    contexts = {i: {"x": i} for i in range(10)}
    np.random.seed(0)
    a = run(RandomSelector(contexts=contexts), 20)["context_id"]
    np.random.seed(0)
    b = [int(np.random.choice(list(np.arange(10)))) for _ in range(20)]
    assert a == b and len(set(a)) > 3


def test_round_robin_stride():
    s = RoundRobinSelector(contexts={i: {} for i in range(5)}, stride=2)
    assert run(s, 6)["context_id"] == [0, 2, 4, 1, 3, 0]
    assert issubclass(RoundRobinSelector, AbstractSelector)
