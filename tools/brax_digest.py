"""Digest of what the Brax kernels compute, for A/B runs of two builds of the library on one box: every family x
a few lane-group widths, seeded contexts and actions, per-call steps and one fused rollout; prints one sha256 per
(family, width) over the final state (float64 view), every observation, reward and flag.  Two builds whose
arithmetic is the same print the same lines.
    CARL_AMD_LIB_PATH=gpurun_in/libcarl_X.so python tools/brax_digest.py [--dump X.npz] > gpurun_out/digest_X.txt
    python tools/brax_digest.py --compare A.npz B.npz      (after 12 + 8 env steps of divergence, if any)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from carl_amd import envs as E  # noqa: E402
from carl_amd.brax_engine import BraxVecEngine  # noqa: E402
from carl_amd.envs.brax.models import SYSTEMS  # noqa: E402

SEL_STATIC = 0
FAMS = ["CARLBraxAnt", "CARLBraxHalfcheetahStiffness", "CARLBraxHumanoidStiffness", "CARLBraxHopper", "CARLBraxWalker2d",
        "CARLBraxInvertedPendulum", "CARLBraxInvertedDoublePendulum", "CARLBraxHumanoidStandup", "CARLBraxReacher",
        "CARLBraxPusher"]


def digest(cls_name, n, width, steps=12, T=8):
    cls = getattr(E, cls_name)
    feats = cls.get_context_features()
    names = list(feats)
    rng = np.random.default_rng(5)
    rows = np.tile([float(f.default_value) for f in feats.values()], (n, 1))
    for k, (lo, hi) in {"gravity": (-15, -5), "friction": (0.3, 1.5), "joint_stiffness": (0.5, 2.0)}.items():
        if k in names:
            rows[:, names.index(k)] = rng.uniform(lo, hi, n)
    rows = rows.astype(np.float32).astype(np.float64)
    s = SYSTEMS[cls.env_name](names)
    if width:
        s.lanes_per_env = width
    eng = BraxVecEngine(s, len(names), rows, n, "cuda", selector=SEL_STATIC, ctx_idx0=np.arange(n), seed=3,
                        max_episode_steps=40, branch_record=True)
    eng.reset()
    h = hashlib.sha256()
    amp = float(max(s.act_hi[: s.n_act]))
    for t in range(steps):
        a = rng.uniform(-amp, amp, (n, s.n_act)).astype(np.float32)
        obs, rew, term, trunc = eng.step(torch.as_tensor(a))
        for x in (obs, rew, term, trunc, eng.branch_sig):
            h.update(np.ascontiguousarray(x.cpu().numpy()).tobytes())
    acts = torch.as_tensor(rng.uniform(-amp, amp, (T, n, s.n_act)).astype(np.float32), device="cuda")
    buf = eng.alloc_rollout(T)
    out = eng.rollout(acts, out=buf)
    torch.cuda.synchronize()
    for k in sorted(out):
        if isinstance(out[k], torch.Tensor):
            h.update(np.ascontiguousarray(out[k].cpu().numpy()).tobytes())
    st = eng.state_np()
    h.update(np.ascontiguousarray(st + 0.0).tobytes())  # (+ 0.0: -0.0 and 0.0 hash alike)
    if DUMP is not None:
        DUMP[f"{cls_name}_{n}_{width}"] = st
    return h.hexdigest()[:16], bool(np.isfinite(st).all())


DUMP = None

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--compare":  # two dumps: the largest difference of the final states
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        for k in a.files:
            d = np.abs(a[k] - b[k]) / (1.0 + np.abs(a[k]))
            print(f"{k:44s} max rel diff {d.max():.2e}  entries differing {np.mean(a[k] != b[k]):.4f}")
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        DUMP = {}
    for f in FAMS:
        for n, w in ((1024, 0), (200, 16), (40000, 0)):
            try:
                d, ok = digest(f, n, w)
                print(f"{f:32s} n {n:5d} width {w:2d}  {d}  finite {ok}", flush=True)
            except Exception as e:  # a width a model does not support
                print(f"{f:32s} n {n:5d} width {w:2d}  -- {type(e).__name__}: {str(e)[:80]}", flush=True)
    if DUMP is not None:
        np.savez(sys.argv[2], **DUMP)
