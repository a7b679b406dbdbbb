"""Golden fixture: the context-feature tables (name, kind, bounds / choices, default) of the reference's
classic-control and Brax env classes (SURVEY.md section 8a rows E-*, a11).

Run in the build container, where /root/reference exists:  python tests/golden/make_feature_table_golden.py
The reference classes cannot be imported here (gymnasium / brax are not installed), so the tables are
read from the class files' syntax trees: every ``key: <Kind>ContextFeature(name, ...)`` entry of the dict
``get_context_features`` returns is evaluated with ``np`` in scope.  The output holds values only."""
from __future__ import annotations

import ast
import json
import math
import os

import numpy as np

REF = "/root/reference/carl/envs"
FILES = {
    "classic_control": os.path.join(REF, "gymnasium", "classic_control"),
    "brax": os.path.join(REF, "brax"),
}


def _num(v):
    if isinstance(v, (list, tuple)):
        return [_num(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _num(x) for k, x in v.items()}
    if isinstance(v, (int, float, np.floating, np.integer)):
        v = float(v)
        if math.isinf(v):
            return "inf" if v > 0 else "-inf"
        return v
    return v


def _goal_wrapper_literal(name: str):
    """a module-level literal of brax_walker_goal_wrapper.py (``directions``: the compass codes the walker
    classes import; ``DIRECTION_NAMES``: their spoken names)"""
    tree = ast.parse(open(os.path.join(REF, "brax", "brax_walker_goal_wrapper.py")).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == name:
            return ast.literal_eval(node.value)
    raise RuntimeError(f"{name} not found")


def _directions() -> list:
    return _goal_wrapper_literal("directions")


def _goal_sentences() -> list:
    """outputs of the reference's ``BraxLanguageWrapper.get_goal_desc`` (the method alone is compiled from the
    module's syntax tree -- the module itself imports gym and brax -- and run on a few contexts)"""
    path = os.path.join(REF, "brax", "brax_walker_goal_wrapper.py")
    tree = ast.parse(open(path).read())
    fn = next(n for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "BraxLanguageWrapper"
              for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "get_goal_desc")
    ns = {"DIRECTION_NAMES": _goal_wrapper_literal("DIRECTION_NAMES")}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    contexts = [{"target_distance": 9.8, "target_direction": 112, "target_radius": 5},
                {"target_distance": 11.764052345967665, "target_direction": 34, "target_radius": 0.5},
                {"target_distance": 100, "target_direction": 1},
                {"target_distance": 2.5, "target_direction": 434}]
    return [{"context": c, "sentence": ns["get_goal_desc"](None, c)} for c in contexts]


def tables_of(path: str) -> dict:
    tree = ast.parse(open(path).read())
    out = {}
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef)]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "get_context_features"]:
            feats = []
            # the returned dict literal: key -> <Kind>ContextFeature(name, ...) in source order (the key is the
            # column name of the context table; the feature's own `name` argument is recorded beside it)
            dicts = [n for n in ast.walk(fn) if isinstance(n, ast.Dict)]
            table = max(dicts, key=lambda d: len(d.keys))
            env = {"np": np, "directions": _directions()}
            for key, call in zip(table.keys, table.values):
                kind = getattr(call.func, "id", getattr(call.func, "attr", ""))
                assert kind.endswith("ContextFeature"), (path, kind)
                args = [eval(compile(ast.Expression(a), path, "eval"), env) for a in call.args]
                kw = {k.arg: eval(compile(ast.Expression(k.value), path, "eval"), env) for k in call.keywords}
                feats.append({"key": ast.literal_eval(key), "kind": kind, "name": args[0], "args": _num(args[1:]),
                              **{k: _num(v) for k, v in kw.items()}})
            if feats:
                out[cls.name] = feats
    return out


def main() -> None:
    golden = {}
    for group, d in FILES.items():
        for f in sorted(os.listdir(d)):
            if f.startswith("carl_") and f.endswith(".py"):
                for cls, feats in tables_of(os.path.join(d, f)).items():
                    golden[cls] = feats
    golden["_goal_wrapper"] = {"sentences": _goal_sentences(), "directions": _goal_wrapper_literal("directions"),
                               "DIRECTION_NAMES": {str(k): v for k, v in _goal_wrapper_literal("DIRECTION_NAMES").items()}}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "context_feature_tables.json")
    with open(dst, "w") as fh:
        json.dump(golden, fh, indent=1, sort_keys=True)
    print(dst, {k: len(v) for k, v in golden.items()})


if __name__ == "__main__":
    main()
