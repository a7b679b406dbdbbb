import sys; sys.path.insert(0, ".")
import numpy as np, torch
from tests.test_gpu_brax_invariants import _make, _bodies, _qrot
from carl_amd import envs as E
name = sys.argv[1] if len(sys.argv) > 1 else "ant"
cls = {"ant": E.CARLBraxAnt, "humanoid": E.CARLBraxHumanoid, "halfcheetah": E.CARLBraxHalfcheetahStiffness}[name]
rng = np.random.default_rng(4)
def rows_fn(rows, names):
    n = len(rows)
    rows[:, names.index("gravity")] = rng.uniform(-15, -5, n)
    if name != "halfcheetah":
        rows[:, names.index("mass_torso")] = rng.uniform(5, 15, n)
        rows[:, names.index("friction")] = rng.uniform(0.3, 1.5, n)
n = 32768
dev = torch.device("cuda", 0)
eng, s, rows, names = _make(cls, n, dev, rows_fn, seed=2, auto_reset=True)
eng.reset()
L = s.n_links
com = torch.tensor([[s.com[i][k] for k in range(3)] for i in range(L)], device=dev)
cl = torch.tensor([s.coll_link[k] for k in range(s.n_coll)], device=dev, dtype=torch.long)
cp = torch.tensor([[s.coll_pos[k][j] for j in range(3)] for k in range(s.n_coll)], device=dev)
cr = torch.tensor([s.coll_radius[k] for k in range(s.n_coll)], device=dev)
g = torch.Generator(device=dev).manual_seed(0)
lo, hi = float(min(s.act_lo[: s.n_act])), float(max(s.act_hi[: s.n_act]))
print("radii", sorted(set(cr.tolist())), "act range", lo, hi, "gear", [s.act_gear[k] for k in range(s.n_act)][:3])
for t in range(120):
    a = torch.rand((n, s.n_act), generator=g, device=dev) * (hi - lo) + lo
    obs, rew, term, trunc = eng.step(a)
    if t % 20 != 19: continue
    p, r, v, w = _bodies(eng, L)
    ctr = p[:, cl] + _qrot(r[:, cl], (cp - com[cl])[None])
    depth = (cr[None] - ctr[..., 2])
    d_env, k_env = depth.max(1)
    q = torch.quantile(d_env, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev))
    print(f"t={t} depth[m] quantiles 50/90/99/99.9/max:", [f"{x:.4f}" for x in q.tolist()], "elapsed of worst", int(eng.elapsed[d_env.argmax()]))
    i = int(d_env.argmax())
    print("   worst env", i, "sphere", int(k_env[i]), "link", int(cl[k_env[i]]), "radius %.3f" % float(cr[k_env[i]]), "ctx", {nm: round(float(rows[i, names.index(nm)]), 2) for nm in ("gravity", "mass_torso", "friction") if nm in names},
          "torso z %.3f" % float(p[i, 0, 2]), "|v| max %.2f |w| max %.2f" % (float(v[i].norm(dim=-1).max()), float(w[i].norm(dim=-1).max())))

# ---- joint gaps vs context
from tests.test_gpu_brax_invariants import _np_qrot
joints = [i for i in range(L) if not (s.parent[i] < 0 and s.n_link_dof[i] == 6) and s.n_slide[i] == 0 and s.parent[i] >= 0]
ac = torch.tensor([[s.joint_pos[i][k] - s.com[i][k] for k in range(3)] for i in joints], device=dev)
ap = torch.tensor(np.array([np.array([s.link_pos[i][k] for k in range(3)]) + _np_qrot([s.link_rot[i][k] for k in range(4)], [s.joint_pos[i][k] for k in range(3)]) - np.array([s.com[s.parent[i]][k] for k in range(3)]) for i in joints]), device=dev, dtype=torch.float32)
ji = torch.tensor(joints, device=dev); pi = torch.tensor([s.parent[i] for i in joints], device=dev)
worst = torch.zeros(n, device=dev)
for t in range(200):
    a = torch.rand((n, s.n_act), generator=g, device=dev) * (hi - lo) + lo
    eng.step(a)
    p, r, v, w = _bodies(eng, L)
    gap = ((p[:, pi] + _qrot(r[:, pi], ap[None])) - (p[:, ji] + _qrot(r[:, ji], ac[None]))).norm(dim=-1).max(1).values
    gap = torch.where(eng.elapsed >= 10, gap, torch.zeros_like(gap))
    worst = torch.maximum(worst, gap)
q = torch.quantile(worst, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev))
print("worst joint gap per env over 200 steps, quantiles 50/90/99/99.9/max [m]:", [f"{x:.4f}" for x in q.tolist()])
if "mass_torso" in names:
    mt = torch.tensor(rows[:, names.index("mass_torso")], device=dev)
    for lo_, hi_ in [(5, 6), (6, 7), (7, 8), (8, 10), (10, 12), (12, 15)]:
        sel = (mt >= lo_) & (mt < hi_)
        print(f"  mass_torso [{lo_},{hi_}): median worst gap {float(worst[sel].median()):.4f}  p99 {float(worst[sel].quantile(0.99)):.4f} max {float(worst[sel].max()):.4f}")
