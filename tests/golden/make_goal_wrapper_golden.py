"""Golden fixture: the arithmetic of the reference's BraxWalkerGoalWrapper (reset + step,
carl/envs/brax/brax_walker_goal_wrapper.py:113-140), produced by RUNNING the reference's own methods.

Run in the build container:  python tests/golden/make_goal_wrapper_golden.py
The module imports gym and brax (absent here), so ``reset`` / ``step`` and the ``direction_values`` /
``STATE_INDICES`` literals are taken from its syntax tree and executed against a stub env that replays
scripted observations.  ``dt`` is an INPUT here (the reference reads the MJCF time step through
brax.io.mjcf, which is unavailable): the fixture pins position integration, progress reward and the
success rule, not the time step."""
from __future__ import annotations

import ast
import json
import os

import numpy as np

PATH = "/root/reference/carl/envs/brax/brax_walker_goal_wrapper.py"


def main() -> None:
    tree = ast.parse(open(PATH).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BraxWalkerGoalWrapper")
    methods = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    state_indices = next(ast.literal_eval(n.value) for n in tree.body
                         if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "STATE_INDICES")
    dv_node = next(n.value for n in ast.walk(methods["__init__"])
                   if isinstance(n, ast.Assign) and getattr(n.targets[0], "attr", "") == "direction_values")
    direction_values = eval(compile(ast.Expression(dv_node), PATH, "eval"), {"np": np})
    ns = {"np": np, "STATE_INDICES": state_indices}
    exec(compile(ast.Module(body=[methods["reset"], methods["step"]], type_ignores=[]), PATH, "exec"), ns)

    class ReplayEnv:
        def __init__(self, seq):
            self.seq, self.t = seq, 0

        def reset(self, seed=None, options=None):
            self.t = 0
            return self.seq[0], {}

        def step(self, action):
            self.t += 1
            return self.seq[self.t], 0.0, False, False, {}

    class Wrapper:
        pass

    rng = np.random.default_rng(0)
    obs_dim = {"ant": 27, "humanoid": 244, "halfcheetah": 18, "hopper": 11, "walker2d": 17}
    cases = []
    for env_name, idx in state_indices.items():
        for code, dist, radius, dt, speed in ((112, 9.8, 5.0, 0.01, 3.0), (34, 0.6, 0.5, 0.003, 40.0), (2, 0.02, 0.05, 0.002, 5.0)):
            T = 12
            # float32-exact observations (the engine's observation dtype); velocities mostly towards the goal
            seq = rng.normal(0.0, 1.0, (T + 1, obs_dim[env_name])).astype(np.float32)
            d = np.asarray(direction_values[code])
            seq[:, idx[0]] = (d[0] * speed + rng.normal(0, 0.3 * speed, T + 1)).astype(np.float32)
            seq[:, idx[1]] = (d[1] * speed + rng.normal(0, 0.3 * speed, T + 1)).astype(np.float32)
            w = Wrapper()
            w.env, w.env_name, w.dt = ReplayEnv([row.astype(np.float64) for row in seq]), env_name, dt
            w.direction_values = direction_values
            w.context = {"target_direction": code, "target_distance": dist, "target_radius": radius}
            w.position = w.goal_position = w.goal_radius = None
            _, info = ns["reset"](w)
            assert info["success"] == 0
            rewards, success, terminated, positions = [], [], [], []
            for _ in range(T):
                _, r, te, _, info = ns["step"](w, None)
                rewards.append(float(r))
                success.append(int(info["success"]))
                terminated.append(bool(te))
                positions.append([float(w.position[0]), float(w.position[1])])
            cases.append({"env_name": env_name, "obs_indices": list(idx), "target_direction": code, "target_distance": dist,
                          "target_radius": radius, "dt": dt, "goal_position": [float(x) for x in w.goal_position],
                          "velocities": [[float(r[idx[0]]), float(r[idx[1]])] for r in seq[1:]],
                          "reward": rewards, "success": success, "terminated": terminated, "position": positions})
    golden = {"STATE_INDICES": state_indices,
              "direction_values": {str(k): [float(v[0]), float(v[1])] for k, v in direction_values.items()},
              "cases": cases}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "goal_wrapper_sequences.json")
    with open(dst, "w") as fh:
        json.dump(golden, fh, indent=1)
    print(dst, len(cases), "cases;", sum(sum(c["success"]) for c in cases), "successful steps")


if __name__ == "__main__":
    main()
