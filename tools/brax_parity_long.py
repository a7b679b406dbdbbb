"""Branch-aware Brax parity over MILLIONS of lane-steps: the HIP kernel steps a large batch with auto-reset on; every
env step is re-computed by the fp64 restatement (oracle/brax_spring.c) from the engine's own state, the oracle work spread
over the host cores (one process per lane chunk).  Prints, per family, the MAXIMUM of |d| / (1 + |x|) over observation
entries and reward on the lane-steps whose contact record and `terminated` flag agree, and the excluded shares.
    python tools/brax_parity_long.py [n_envs] [steps] [family ...]        (on the GPU box)"""
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

FAMS = {"ant": ("CARLBraxAnt", {"gravity": (-15, -5), "friction": (0.3, 1.5), "mass_torso": (5, 15)}, 1.2),
        "halfcheetah": ("CARLBraxHalfcheetahStiffness", {"joint_stiffness": (0.5, 2.0), "gravity": (-15, -5)}, 1.2),
        "humanoid": ("CARLBraxHumanoidStiffness", {"joint_stiffness": (0.5, 2.0), "gravity": (-15, -5)}, 0.48),
        "hopper": ("CARLBraxHopper", {"gravity": (-15, -5)}, 1.2), "walker2d": ("CARLBraxWalker2d", {"gravity": (-15, -5)}, 1.2)}
_W = {}


def _init(fam):
    from carl_amd import envs as E
    from carl_amd.envs.brax.models import SYSTEMS

    cls = getattr(E, FAMS[fam][0])
    _W["names"] = list(cls.get_context_features())
    _W["sys"] = SYSTEMS[cls.env_name](_W["names"])


def _work(job):
    from oracle import brax as B
    from oracle import oracle as O

    rows, state, action, elapsed = job
    n = len(rows)
    eng = B.Engine(_W["sys"], rows, n, selector=O.SEL_STATIC, ctx_idx0=np.arange(n), autoreset=False, max_steps=1 << 30)
    eng.state[:] = state
    eng.elapsed[:] = elapsed
    out = eng.step(action)
    return out.obs, out.reward, out.terminated, eng.branch_sig


def run(fam, n, steps, pool, n_chunks):
    import torch

    from carl_amd import envs as E
    from carl_amd.brax_engine import BraxVecEngine
    from carl_amd.envs.brax.models import SYSTEMS
    from oracle import oracle as O

    cls = getattr(E, FAMS[fam][0])
    feats = cls.get_context_features()
    names = list(feats)
    rng = np.random.default_rng(11)
    rows = np.tile([float(f.default_value) for f in feats.values()], (n, 1))
    for k, (lo, hi) in FAMS[fam][1].items():
        rows[:, names.index(k)] = rng.uniform(lo, hi, n)
    rows = rows.astype(np.float32).astype(np.float64)
    s = SYSTEMS[cls.env_name](names)
    eng = BraxVecEngine(s, len(names), rows, n, "cuda", selector=O.SEL_STATIC, ctx_idx0=np.arange(n), seed=3,
                        max_episode_steps=1000, branch_record=True)
    eng.reset()
    amp = FAMS[fam][2] * float(max(s.act_hi[: s.n_act]))
    cuts = np.linspace(0, n, n_chunks + 1).astype(int)
    worst, n_agree, n_contact, n_flag, total, over = 0.0, 0, 0, 0, 0, 0
    col_worst, worst_at = np.zeros(s.obs_dim), None
    for t in range(steps):
        state = eng.state_np()
        elapsed = eng.elapsed.cpu().numpy()
        a = rng.uniform(-amp, amp, (n, s.n_act)).astype(np.float32)
        obs, rew, term, trunc = eng.step(torch.as_tensor(a))
        res = pool.map(_work, [(rows[i:j], state[i:j], a[i:j], elapsed[i:j]) for i, j in zip(cuts[:-1], cuts[1:])])
        o_obs = np.concatenate([r[0] for r in res]); o_rew = np.concatenate([r[1] for r in res])
        o_term = np.concatenate([r[2] for r in res]); o_sig = np.concatenate([r[3] for r in res])
        term_g = term.cpu().numpy() != 0
        done = term_g | (trunc.cpu().numpy() != 0)
        got = np.where(done[:, None], eng.final_obs.cpu().numpy(), obs.cpu().numpy())  # the transition's own observation
        sig = eng.branch_sig.cpu().numpy().view(np.uint32)
        flag = term_g != (o_term != 0)
        contact = (sig[:, 0] != o_sig[:, 0]) & ~flag
        agree = ~flag & ~contact
        eo = np.abs(got.astype(np.float64) - o_obs) / (1 + np.abs(o_obs))
        e = np.maximum(eo.max(1), np.abs(rew.cpu().numpy().astype(np.float64) - o_rew) / (1 + np.abs(o_rew)))
        col_worst = np.maximum(col_worst, eo[agree].max(0)) if agree.any() else col_worst
        if agree.any() and float(e[agree].max()) > worst:  # where the run's worst entry sits (column, value, step, env state)
            k = int(np.argmax(np.where(agree, e, -1.0)))
            c = int(np.argmax(eo[k]))
            worst_at = (t, k, c, float(o_obs[k, c]), float(got[k, c]), int(elapsed[k]), bool(done[k]))
        worst = max(worst, float(e[agree].max()))
        over += int((e[agree] > 1e-5).sum())
        n_agree += int(agree.sum()); n_contact += int(contact.sum()); n_flag += int(flag.sum()); total += n
    print(f"{fam:12s} {total:9d} lane-steps ({n} envs x {steps} steps, auto-reset on, {int(eng.episodes_done.sum())} episodes): agreeing lanes max {worst:.2e}, "
          f"above 1e-5: {over}; excluded: contact record differs {n_contact} ({n_contact / total:.2e}), terminated differs {n_flag} ({n_flag / total:.2e})", flush=True)
    if os.environ.get("CARL_PARITY_DETAIL") == "1":
        top = np.argsort(-col_worst)[:10]
        print("   worst entry (step, lane, column, oracle value, engine value, elapsed, done):", worst_at)
        print("   worst columns:", [(int(c), f"{col_worst[c]:.2e}") for c in top], flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    fams = sys.argv[3:] or ["ant", "halfcheetah", "humanoid"]
    procs = min(os.cpu_count() or 8, 96)
    ctx = mp.get_context("spawn")
    for fam in fams:
        with ctx.Pool(procs, initializer=_init, initargs=(fam,)) as pool:
            run(fam, n, steps, pool, procs)
