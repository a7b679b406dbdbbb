"""HumanoidStandup under the spring-branch gears (350 / 100): how many envs of a batch ever leave |obs| <= 500 in T steps
of full-range random actions, for a few choices of this build's OWN spring constants (rows marked H).
    python tools/diag_standup_sweep.py [n_envs] [steps]      (GPU box)"""
import sys

import torch

sys.path.insert(0, ".")
from carl_amd import envs as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
MJCF = [100, 100, 100, 100, 100, 300, 200, 100, 100, 300, 200, 25, 25, 25, 25, 25, 25]
CASES = [("product table", {}),
         ("MJCF gears", {"gear": MJCF}),
         ("k_vel 200", {"k_vel": 200.0}),
         ("k_ang_damp 40", {"k_ang_damp": 40.0}),
         ("k_limit 2500", {"k_limit": 2500.0}),
         ("k_pos 27000 k_vel 80 k_limit 2500 k_ang 30", {"k_pos": 27000.0, "k_vel": 80.0, "k_limit": 2500.0, "k_ang_damp": 30.0}),
         ("k_pos 10000", {"k_pos": 10000.0}),
         ("erp 0.05", {"erp": 0.05}),
         ("arms 50", {"gear": [350] * 11 + [50] * 6}),
         ("legs 350 arms 25", {"gear": [350] * 11 + [25] * 6})]
for cls in (E.CARLBraxHumanoidStandup, E.CARLBraxHumanoid):
    for label, ch in CASES:
        env = cls(batch_size=n, device="cuda:0")
        eng = env.env
        s = eng.sys
        for k, g in enumerate(ch.get("gear", [])):
            s.act_gear[k] = float(g)
        for i in range(s.n_links):
            for f in ("k_pos", "k_vel", "k_limit", "k_ang_damp"):
                if f in ch:
                    getattr(s, f)[i] = ch[f]
        if "erp" in ch:
            s.baumgarte_erp = ch["erp"]
        eng.sys_dev = torch.frombuffer(bytearray(bytes(s)), dtype=torch.uint8).to(eng.device)
        env.reset(seed=0)
        chunk = 50
        out = eng.alloc_rollout(chunk)
        g = torch.Generator(device="cuda:0").manual_seed(0)
        bad = torch.zeros(n, dtype=torch.bool, device="cuda:0")
        first = None
        for k in range(T // chunk):
            a = torch.rand((chunk, n, s.n_act), device="cuda:0", generator=g) * 0.8 - 0.4
            eng.rollout(a, out)
            o = out["obs"]
            b = (~torch.isfinite(o)).any(dim=2) | (o.abs().amax(dim=2) > 500)
            if first is None and bool(b.any()):
                first = k * chunk + int(b.any(dim=1).nonzero()[0])
            bad |= b.any(dim=0)
        print(f"{cls.__name__:24s} {label:46s} envs ever above 500 / non-finite: {int(bad.sum()):5d} of {n}   first at step {first}")
