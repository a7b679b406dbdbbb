"""Generate tests/golden/selector_sequences.json by RUNNING the reference's selector
module (the only hot-path-adjacent reference code importable in the build container:
carl/context/selection.py needs numpy only).

    PYTHONPATH=/root/reference python tests/golden/make_selector_golden.py

The output is data (id / n_calls / key sequences), not reference source.  /root/reference
does not exist on the GPU box; tests read only the committed JSON.
"""
import json
import os
import sys

sys.path.insert(0, "/root/reference")
from carl.context.selection import (  # noqa: E402
    CustomSelector,
    RoundRobinSelector,
    StaticSelector,
)

out = {}


def run(selector, n):
    ids, calls, keys = [], [], []
    for _ in range(n):
        selector.select()
        ids.append(int(selector.context_id))
        calls.append(int(selector.n_calls))
        keys.append(selector.context_key)
    return {"context_id": ids, "n_calls": calls, "context_key": keys}


for n_ctx in (1, 2, 3, 5, 7):
    contexts = {chr(ord("a") + i): {"x": float(i)} for i in range(n_ctx)}
    out[f"round_robin_{n_ctx}"] = run(RoundRobinSelector(contexts=contexts), 12)
    out[f"static_{n_ctx}"] = run(StaticSelector(contexts=contexts), 6)


def fn(inst):
    cid = 1 if inst.n_calls == 0 else 0
    return inst.contexts[inst.contexts_keys[cid]], cid


contexts = {k: {"x": 0.0} for k in "abc"}
out["custom_first1_then0"] = run(CustomSelector(contexts=contexts, selector_function=fn), 5)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "selector_sequences.json")
with open(path, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("wrote", path)
