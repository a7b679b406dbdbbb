"""Brax one-step parity over WIDER context ranges than BASELINE's configs (run on the GPU box; a checker's tool: imports oracle/).

    python tools/fuzz_brax_contexts.py [--n=16384] [--steps=30]

Ant / Halfcheetah / Humanoid, every context row drawn wide inside the reference's declared bounds (carl_ant.py:21-49 and the
like): gravity log-uniform in [-50, -2], friction in [0.1, 10], elasticity U(0, 0.8), mass_torso x [0.5, 3] (above this
build's stability floors), joint_stiffness x [0.3, 3] where the family has it.  The engine free-runs `steps` env steps from reset
under random actions; the float64 restatement then re-computes ONE env step of every lane from the engine's own state: worst
|d| / (1 + |x|) over the lanes whose contact / limit record agrees, the excluded share, and non-finite lanes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import brax as B  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.test_gpu_brax import rel_err  # noqa: E402


def logu(rng, lo, hi, n):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), n))


def widen(rng, n, default, names, fam, heavy_only):
    from carl_amd.envs.brax.feature_tables import JOINT_STIFFNESS_CEILING

    rows = np.tile(default, (n, 1))
    rows[:, names.index("gravity")] = -logu(rng, 2.0, 50.0, n)
    rows[:, names.index("friction")] = logu(rng, 0.1, 10.0, n)
    rows[:, names.index("elasticity")] = rng.uniform(0.0, 0.8, n)
    if "ang_damping" in names:
        rows[:, names.index("ang_damping")] = -rng.uniform(0.0, 0.5, n)
    for k, nm in enumerate(names):  # every link mass from its default upwards (lighter links: the documented stability floors)
        if nm.startswith("mass_"):
            rows[:, k] = default[k] * logu(rng, 1.0 if heavy_only else 0.5, 3.0, n)
    if "joint_stiffness" in names:
        rows[:, names.index("joint_stiffness")] = logu(rng, 0.3, JOINT_STIFFNESS_CEILING.get(fam, 2.0), n)
    return rows.astype(np.float32).astype(np.float64)


def main():
    import inspect

    import carl_amd.envs.brax as brax_envs
    from carl_amd.brax_engine import BraxVecEngine
    from carl_amd.envs.brax.models import SYSTEMS

    n, steps = 8192, 60
    for a in sys.argv[1:]:
        if a.startswith("--n="):
            n = int(a.split("=")[1])
        if a.startswith("--steps="):
            steps = int(a.split("=")[1])
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2026)
    for cname, cls in inspect.getmembers(brax_envs):
        if not (inspect.isclass(cls) and cname.startswith("CARLBrax") and cname != "CARLBraxEnv"):
            continue
        feats = cls.get_context_features()
        names = list(feats)
        default = np.array([float(f.default_value) for f in feats.values()])
        fam = cls.env_name
        s = SYSTEMS[fam](names)
        amp = 0.4 if "humanoid" in fam else 1.0
        rows = widen(rng, n, default, names, fam, heavy_only=True)
        eng = BraxVecEngine(s, len(names), rows, n, dev, selector=O.SEL_STATIC, seed=1, ctx_idx0=np.arange(n), auto_reset=False,
                            max_episode_steps=10_000, branch_record=True)
        eng.reset()
        g = torch.Generator(device=dev).manual_seed(3)
        for _ in range(steps):
            eng.step((torch.rand((n, s.n_act), generator=g, device=dev) * 2 - 1) * amp)
        st = eng.state_np()
        fin0 = np.isfinite(st.reshape(n, -1)).all(1)
        ora = B.Engine(s, rows, n, selector=O.SEL_STATIC, ctx_idx0=np.arange(n), autoreset=False, max_steps=10_000)
        ora.reset()
        ora.state[:] = np.where(fin0.reshape((n,) + (1,) * (st.ndim - 1)), st, ora.state)
        ora.elapsed[:] = eng.elapsed.cpu().numpy()
        a = (rng.uniform(-1, 1, (n, s.n_act)) * amp).astype(np.float32)
        obs, rew, term, trunc = eng.step(torch.as_tensor(a))
        out = ora.step(a)
        sig = eng.branch_sig.cpu().numpy().view(np.uint32)
        o = obs.cpu().numpy()
        fin = fin0 & np.isfinite(o).all(1) & np.isfinite(out.obs).all(1) & (np.abs(o).max(1) < 1e4)
        flag = (term.cpu().numpy() != 0) != (out.terminated != 0)
        agree = (sig[:, 0] == ora.branch_sig[:, 0]) & ~flag & fin
        e = np.where(fin, np.maximum(rel_err(o, out.obs).max(1), rel_err(rew.cpu().numpy(), out.reward)), 0.0)
        k = int(np.argmax(np.where(agree, e, -1)))
        above = agree & (e > 1e-5)
        print(f"{cname:32s} {n} lanes after {steps} free-running steps: agreeing {int(agree.sum())} worst {e[k]:.2e}, above 1e-5: {int(above.sum())}; "
              f"excluded (contact record / flag differs) {int((~agree & fin).sum())}; blown up (non-finite or |obs| > 1e4) {int((~fin).sum())}")
        badl = above | ~fin
        if badl.any():  # which feature separates the lanes that left the bar (or blew up) from the rest?
            for nm in names:
                c = rows[:, names.index(nm)]
                if np.ptp(c) == 0:
                    continue
                q = lambda v: np.array2string(np.quantile(v, [0, 0.1, 0.5, 0.9, 1]), precision=3)
                print(f"    {nm:28s} bad lanes quantiles {q(c[badl])}   good lanes {q(c[~badl])}")


if __name__ == "__main__":
    main()
