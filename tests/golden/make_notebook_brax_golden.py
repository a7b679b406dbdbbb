"""Extracts the one Brax transition the reference records: examples/brax_with_goals.ipynb, cell 3
prints the observation and reward of CARLBraxAnt (use_language_goals=True) after reset() + ONE
step with ``env.action_space.sample()`` (action and JAX PRNG state are not recorded).  It cannot
pin the arithmetic, but it is real reference output: it pins the observation layout (z, quaternion,
8 joint angles in actuator-independent q order, 14 velocities), the initial pose, the reset-noise
scale and the velocity scale after one control step, and the goal wrapper's reward for a step that
moves away from / barely towards the goal.  Run from the repo root with /root/reference present:
    python tests/golden/make_notebook_brax_golden.py
"""
import json
import re

nb = json.load(open("/root/reference/examples/brax_with_goals.ipynb"))
cell = nb["cells"][3]
text = "".join(o.get("text", "") if isinstance(o.get("text", ""), str) else "".join(o["text"]) for o in cell["outputs"]
               if "text" in o)
arr = re.search(r"Array\(\[(.*?)\], dtype=float32\)", text, re.S).group(1)
obs = [float(x) for x in arr.replace("\n", " ").split(",")]
ctx = re.search(r"'context': (\{.*?\})\}", text, re.S).group(1)
context = json.loads(ctx.replace("'", '"'))
reward = float(text.strip().splitlines()[-1])
assert len(obs) == 27
json.dump({"source": "examples/brax_with_goals.ipynb cell 3 (CARLBraxAnt, reset + 1 random step)",
           "obs": obs, "reward": reward, "context": context},
          open("tests/golden/notebook_brax_ant_first_step.json", "w"), indent=1)
print(obs[:5], reward, context)
