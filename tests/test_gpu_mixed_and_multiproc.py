"""Mixed-family batches (BASELINE configs 3 and 5 as ONE object), the replayable per-call step, and the
HIP engine under a process group on real hardware -- needs an MI355X.

Everything here is a bit-exactness statement: a mixed batch, a captured hipGraph step, or a lane shard run
by another process must produce exactly the bytes the plain engine produces.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.test_gpu_parity import _engine, random_actions, random_table

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_BOOKKEEPING = ("state", "elapsed", "ctx_idx", "episode", "n_calls", "ep_return", "last_return", "last_length",
                "episodes_done")


def _pair(device, n, seed=3, **kw):
    rng = np.random.default_rng(seed)
    fams = (O.ACROBOT, O.MOUNTAINCAR)
    tables = [random_table(f, rng, n) for f in fams]
    mk = lambda: [_engine(f, t, n, device, selector=O.SEL_STATIC, seed=5, ctx_idx0=np.arange(n),  # noqa: E731
                          lane_offset=k * n, **kw) for k, (f, t) in enumerate(zip(fams, tables))]
    return fams, rng, mk


@pytest.mark.parametrize("n,T", [(4096, 260), (65536, 250)])
def test_mixed_batch_equals_separate_engines(n, T, device):
    """BASELINE config 3 (Acrobot + MountainCar, 65 536 contexts each) as one MixedVecEngine: both families'
    launches of a fused rollout run on their own streams, joined into the caller's; every output and every
    counter equals the two engines run one after the other; the episodic returns are ONE [2n] vector."""
    from carl_amd.mixed import MixedVecEngine

    fams, rng, mk = _pair(device, n)
    sep, parts = mk(), mk()
    mixed = MixedVecEngine(parts, ["acrobot", "mountaincar"])
    acts = [torch.as_tensor(random_actions(f, rng, (T, n)), device=device) for f in fams]
    for e in sep:
        e.reset()
    mixed.reset()
    outs = mixed.rollout(acts)
    ref = [e.rollout(a) for e, a in zip(sep, acts)]
    torch.cuda.synchronize()
    for k in range(2):
        for name in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(outs[k][name], ref[k][name]), (k, name)
        for name in _BOOKKEEPING:
            assert torch.equal(getattr(parts[k], name), getattr(sep[k], name)), (k, name)
    assert mixed.last_return.shape == (2 * n,) and mixed.last_return.is_contiguous()
    assert torch.equal(mixed.last_return, torch.cat([e.last_return for e in sep]))
    assert torch.equal(mixed.episodes_done, torch.cat([e.episodes_done for e in sep]))
    # MountainCar truncates at 200 (Acrobot at 500): every MountainCar lane finished an episode
    assert int(mixed.episodes_done[mixed.part_slice(1)].min()) >= 1


@pytest.mark.parametrize("other", [O.MOUNTAINCAR, O.CARTPOLE, O.PENDULUM, O.MOUNTAINCAR_CONT], ids=lambda f: O.FAMILY_NAMES[f])
@pytest.mark.parametrize("n_a,n_b,T,max_steps", [(4112, 1040, 37, 5), (1024, 4096, 64, 0), (1003, 517, 21, 7)])
def test_pair_launch_equals_two_launches_bit_for_bit(other, n_a, n_b, T, max_steps, device):
    """`carl_rollout_pair`: Acrobot + one other family as ONE heterogeneous launch at 4-step chunks ==
    the two families' own `carl_rollout` launches (8-step chunks), in every output and counter: unequal part sizes,
    ragged last workgroups (4112 = 16 x 256 + 16), a ragged last chunk (37 = 9 x 4 + 1), resets of every lane inside
    the window (TimeLimit 5), either order of the parts; lane counts that are not multiples of 16 (1 003 + 517: rows at
    the ABI-9 pitch -- what the uneven shards of a multi-GPU run of BASELINE config 3 look like).  Combinations the
    library declines (int64 actions, terminal observations) fall back to two launches with the same results."""
    from carl_amd.mixed import MixedVecEngine

    rng = np.random.default_rng(T + other)
    fams, sizes = (O.ACROBOT, other), (n_a, n_b)
    tables = [random_table(f, rng, n) for f, n in zip(fams, sizes)]
    kw = dict(selector=O.SEL_STATIC, seed=5)
    if max_steps:
        kw["max_episode_steps"] = max_steps
    mk = lambda order: [_engine(fams[k], tables[k], sizes[k], device, ctx_idx0=np.arange(sizes[k]), lane_offset=k * 8192,  # noqa: E731
                                **kw) for k in order]
    acts = [torch.as_tensor(random_actions(f, rng, (T, n)), device=device) for f, n in zip(fams, sizes)]
    for order in ((0, 1), (1, 0)):
        sep, parts = mk(order), mk(order)
        mixed = MixedVecEngine(parts)
        for e in sep:
            e.reset()
        mixed.reset()
        a = [acts[k] for k in order]
        outs = mixed.rollout(a)
        assert mixed.pair_launches == 1  # one launch for both families
        ref = [e.rollout(x) for e, x in zip(sep, a)]
        for k in range(2):
            for name in ("obs", "reward", "terminated", "truncated"):
                assert torch.equal(outs[k][name], ref[k][name]), (order, k, name)
            for name in _BOOKKEEPING:
                assert torch.equal(getattr(parts[k], name), getattr(sep[k], name)), (order, k, name)
        if max_steps:
            assert int(mixed.episodes_done.min()) >= T // max_steps
        # declined combinations: terminal observations requested; int64 actions (discrete families)
        outs2 = mixed.rollout(a, [p.alloc_rollout(T, final_obs=True) for p in parts])
        ref2 = [e.rollout(x, e.alloc_rollout(T, final_obs=True)) for e, x in zip(sep, a)]
        a64 = [x.long() if x.dtype == torch.int32 else x for x in a]
        outs3 = mixed.rollout(a64)
        ref3 = [e.rollout(x) for e, x in zip(sep, a64)]
        assert mixed.pair_launches == 1
        for k in range(2):
            for name in ("obs", "reward", "terminated", "truncated"):
                assert torch.equal(outs2[k][name], ref2[k][name]) and torch.equal(outs3[k][name], ref3[k][name]), (k, name)
            assert torch.equal(parts[k].state, sep[k].state)


def test_pair_launch_is_refused_for_other_combinations(device):
    """ABI: CARL_ERR_UNSUPPORTED, nothing enqueued, for two float32 families / two Acrobots / a moving selector"""
    import ctypes as C

    from carl_amd import _lib

    n, T = 1024, 8
    rng = np.random.default_rng(0)

    def call(ea, eb):
        acts = [torch.as_tensor(random_actions(e.family, rng, (T, n)), device=device) for e in (ea, eb)]
        ios = []
        for e, a in zip((ea, eb), acts):
            aa, dt = e._action_tensor(a, (T,))
            ios.append(e._rollout_io(aa, dt, e.alloc_rollout(T), T))
        return ea.lib.carl_rollout_pair(C.byref(ea.b), C.byref(ios[0]), C.byref(eb.b), C.byref(ios[1]), T, ea._stream())

    mk = lambda f, **kw: _engine(f, random_table(f, rng, n), n, device, **{"selector": O.SEL_STATIC, **kw})  # noqa: E731
    assert call(mk(O.PENDULUM), mk(O.MOUNTAINCAR)) == _lib.ERR_UNSUPPORTED
    assert call(mk(O.ACROBOT), mk(O.ACROBOT)) == _lib.ERR_UNSUPPORTED
    assert call(mk(O.ACROBOT, selector=O.SEL_ROUND_ROBIN), mk(O.MOUNTAINCAR)) == _lib.ERR_UNSUPPORTED
    assert call(mk(O.ACROBOT, acrobot_fp32=True), mk(O.MOUNTAINCAR)) == _lib.ERR_UNSUPPORTED
    ea, eb = mk(O.ACROBOT), mk(O.MOUNTAINCAR)
    ea.reset()
    eb.reset()
    assert call(ea, eb) == 0 and call(eb, ea) == 0  # either order
    torch.cuda.synchronize()


def test_mixed_batch_per_call_step(device):
    """the per-call path of a mixed batch: reward / flags arrive as one [2n] vector"""
    from carl_amd.mixed import MixedVecEngine

    n, T = 2048, 40
    fams, rng, mk = _pair(device, n, max_episode_steps=7)
    sep, parts = mk(), mk()
    mixed = MixedVecEngine(parts)
    acts = [torch.as_tensor(random_actions(f, rng, (T, n)), device=device) for f in fams]
    for e in sep:
        e.reset()
    mixed.reset()
    for t in range(T):
        obs, rew, term, trunc = mixed.step([a[t] for a in acts])
        for k, e in enumerate(sep):
            o, r, te, tr = e.step(acts[k][t])
            sl = mixed.part_slice(k)
            assert torch.equal(obs[k], o) and torch.equal(rew[sl], r), (t, k)
            assert torch.equal(term[sl], te) and torch.equal(trunc[sl], tr), (t, k)
    assert int(mixed.episodes_done.sum()) >= 2 * n * (T // 7)


def test_mixed_brax_batch_equals_separate_engines(device):
    """BASELINE config 5 (Halfcheetah + Humanoid) as one object, at a size the test finishes in seconds"""
    from carl_amd.envs import CARLBraxHalfcheetah, CARLBraxHumanoid
    from carl_amd.context.selection import StaticSelector
    from carl_amd.mixed import MixedVecEngine

    n, T = 1024, 12
    def mk():
        return [cls(batch_size=n, device=device, context_selector=StaticSelector, seed=2, lane_offset=k * n,
                    autotune=False) for k, cls in enumerate((CARLBraxHalfcheetah, CARLBraxHumanoid))]
    sep, parts = mk(), mk()
    mixed = MixedVecEngine([p.env for p in parts], ["halfcheetah", "humanoid"])
    g = torch.Generator(device=device).manual_seed(0)
    acts = [torch.rand((T, n, e.env.info.action_dim), generator=g, device=device) * 0.8 - 0.4 for e in sep]
    for e in sep:
        e.reset(seed=2)
    for p in parts:
        p.env.seed(2)
    mixed.reset()
    outs = mixed.rollout(acts)
    ref = [e.env.rollout(a) for e, a in zip(sep, acts)]
    for k in range(2):
        for name in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(outs[k][name], ref[k][name]), (k, name)
    assert torch.equal(mixed.ep_return, torch.cat([e.env.ep_return for e in sep]))


@pytest.mark.parametrize("fam", [O.CARTPOLE, O.PENDULUM], ids=["cartpole", "pendulum"])
def test_captured_step_equals_eager_step(fam, device):
    """VecEngine.capture_step: a hipGraph of the per-call launch reading a fixed action buffer; the capture
    leaves the engine untouched and every replay equals an eager step, through auto-resets"""
    n, T = 4096, 60
    rng = np.random.default_rng(fam)
    table = random_table(fam, rng, n)
    acts = torch.as_tensor(random_actions(fam, rng, (T, n)), device=device)
    kw = dict(selector=O.SEL_STATIC, seed=9, ctx_idx0=np.arange(n), max_episode_steps=11)
    e1, e2 = _engine(fam, table, n, device, **kw), _engine(fam, table, n, device, **kw)
    e1.reset()
    e2.reset()
    buf = acts[0].clone()
    before = e1.snapshot()
    g = e1.capture_step(buf)
    for k, v in before.items():
        assert torch.equal(getattr(e1, k), v), k
    for t in range(T):
        buf.copy_(acts[t])
        o1, r1, te1, tr1 = g.replay()
        o2, r2, te2, tr2 = e2.step(acts[t])
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), t
    for name in _BOOKKEEPING:
        assert torch.equal(getattr(e1, name), getattr(e2, name)), name
    assert g.graph is None  # (round 5: ONE step per replay is served by the eager fast path -- a graph launch costs more)
    # the hipGraph form proper: two env steps per replay on the fixed buffer
    e3, e4 = _engine(fam, table, n, device, **kw), _engine(fam, table, n, device, **kw)
    e3.reset()
    e4.reset()
    before = e3.snapshot()
    g2 = e3.capture_step(buf, n_steps=2)
    assert g2.graph is not None
    for k, v in before.items():
        assert torch.equal(getattr(e3, k), v), k
    for t in range(T // 2):
        buf.copy_(acts[t])
        o1, r1, te1, tr1 = g2.replay()
        e4.step(acts[t])
        o2, r2, te2, tr2 = e4.step(acts[t])
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), t
    for name in _BOOKKEEPING:
        assert torch.equal(getattr(e3, name), getattr(e4, name)), name


def test_step_fast_path_tracks_a_changed_action_tensor(device):
    """the per-call fast path skips validation only for the SAME tensor at the same address: a new tensor, a
    NumPy array or another dtype goes through validation again"""
    fam, n = O.MOUNTAINCAR, 512
    e1 = _engine(fam, random_table(fam, np.random.default_rng(0), n), n, device, selector=O.SEL_STATIC, seed=1)
    e2 = _engine(fam, random_table(fam, np.random.default_rng(0), n), n, device, selector=O.SEL_STATIC, seed=1)
    e1.reset(); e2.reset()
    a32 = torch.randint(0, 3, (n,), dtype=torch.int32, device=device)
    a64 = torch.randint(0, 3, (n,), dtype=torch.int64, device=device)
    seq = [a32, a32, a64, a64.cpu().numpy(), a32, a32.clone(), a64]
    for a in seq:
        r1 = e1.step(a)
        r2 = e2.step(torch.as_tensor(a).to(device).clone())
        assert all(torch.equal(x, y) for x, y in zip(r1, r2))
    with pytest.raises(ValueError):
        e1.step(a32[:-1])


_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["CARL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from carl_amd.engine import VecEngine
from carl_amd.distributed import lane_shard, shard_context_rows, all_gather_episode_stats, reduce_episode_summary
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ["CARL_BACKEND"]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
d = np.load(os.environ["CARL_CASE"])
N, T = int(d["acts"].shape[1]), int(d["acts"].shape[0])
sh = lane_shard(N, rank, world)
eng = VecEngine(int(d["fam"]), shard_context_rows(d["table"], sh, True), sh.count, dev, selector=0, seed=13,
                lane_offset=sh.offset, ctx_idx0=np.arange(sh.count), max_episode_steps=int(d["max_steps"]))
eng.reset()
out = eng.rollout(torch.as_tensor(np.ascontiguousarray(d["acts"][:, sh.slice]), device=dev))
torch.cuda.synchronize()
if backend == "nccl":
    stats = all_gather_episode_stats(eng)
    summary = reduce_episode_summary(eng)
else:  # gloo moves host tensors: the HIP engine still ran under the process group, the collective is staged
    stats = all_gather_episode_stats({k: getattr(eng, k).cpu() for k in ("last_return", "last_length", "episodes_done")})
    summary = reduce_episode_summary({k: getattr(eng, k).cpu() for k in ("last_return", "last_length", "episodes_done")})
if rank == 0:
    np.savez(os.environ["CARL_OUT"], backend=backend, summary_mean=summary["mean_return"],
             **{k: v.cpu().numpy() for k, v in stats.items()})
np.save(os.environ["CARL_OUT"] + f".obs{rank}.npy", out["obs"][-1].cpu().numpy())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_hip_engine_under_a_process_group_on_one_gpu(backend, device, tmp_path):
    """Two ranks (two processes) share the ONE visible GPU: each runs the HIP engine on its lane shard and the
    episodic returns are all-gathered -- with RCCL (backend nccl) when it accepts two ranks on one device, and
    with gloo (host-staged) always.  The gathered vectors must equal the single-process engine's bit for bit:
    the HIP engine, not the oracle, under `lane_shard` + `all_gather_episode_stats` on hardware."""
    fam, N, T, max_steps = O.CARTPOLE, 8192 + 48, 96, 17  # uneven split: 4120 / 4120
    rng = np.random.default_rng(21)
    table = random_table(fam, rng, N)
    acts = random_actions(fam, rng, (T, N))
    case = tmp_path / "case.npz"
    np.savez(case, fam=fam, table=table, acts=acts, max_steps=max_steps)
    out = tmp_path / f"out_{backend}.npz"
    port = 29500 + (os.getpid() % 2000) + (7 if backend == "gloo" else 0)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CARL_ROOT=ROOT, CARL_CASE=str(case), CARL_OUT=str(out),
                   CARL_BACKEND=backend, CARL_AMD_NO_BUILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    failed = False
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            failed = True
        logs.append(o)
        failed |= p.returncode != 0
    if failed and backend == "nccl":
        # RCCL refuses (or hangs on) two ranks that map to one device on this build: recorded, not hidden --
        # the gloo variant of this test is the one that must pass
        pytest.skip("RCCL does not run two ranks on one device here: " + " | ".join(l[-300:] for l in logs))
    assert not failed, "\n".join(logs)
    got = np.load(out)
    ref = _engine(fam, table, N, device, selector=0, seed=13, ctx_idx0=np.arange(N), max_episode_steps=max_steps)
    ref.reset()
    ref_out = ref.rollout(torch.as_tensor(acts, device=device))
    for k in ("last_return", "last_length", "episodes_done"):
        np.testing.assert_array_equal(got[k], getattr(ref, k).cpu().numpy(), err_msg=k)
    half = N // 2
    np.testing.assert_array_equal(np.load(str(out) + ".obs0.npy"), ref_out["obs"][-1][:half].cpu().numpy())
    np.testing.assert_array_equal(np.load(str(out) + ".obs1.npy"), ref_out["obs"][-1][half:].cpu().numpy())
    fin = got["episodes_done"] > 0
    assert abs(float(got["summary_mean"]) - float(got["last_return"][fin].mean())) < 1e-4


@pytest.mark.parametrize("selector_name", ["static", "round_robin"])
def test_uneven_lane_shards_through_the_env_api_keep_their_contexts(selector_name, device):
    """10 lanes over 3 ranks (shards 4 / 3 / 3).  Through the CARLEnv constructors -- no explicit
    ctx_idx0 -- every shard must read ITS OWN rows of the context set: with the table sharded like the lanes
    (static selector, lane i <-> context i) and with the whole table replicated (round robin); the shards'
    transitions must equal the unsharded env's, bit for bit."""
    from carl_amd.context.selection import RoundRobinSelector, StaticSelector
    from carl_amd.context.table import ContextTable
    from carl_amd.distributed import lane_shard, shard_context_rows
    from carl_amd.envs import CARLPendulum

    N, T = 10, 30
    rng = np.random.default_rng(2)
    names = list(CARLPendulum.get_context_features())
    rows = random_table(O.PENDULUM, rng, N)
    sel = StaticSelector if selector_name == "static" else RoundRobinSelector
    identity = selector_name == "static"
    acts = torch.as_tensor(random_actions(O.PENDULUM, rng, (T, N)), device=device)

    def run(env, a):
        env.reset(seed=3)
        out = env.env.rollout(a)
        return out["obs"].clone(), out["reward"].clone(), env.env.ctx_idx.clone()

    full = CARLPendulum(contexts=ContextTable(names, rows), num_envs=N, device=device, context_selector=sel,
                        max_episode_steps=7)
    f_obs, f_rew, f_idx = run(full, acts)
    for r in range(3):
        sh = lane_shard(N, r, 3)
        local = shard_context_rows(rows, sh, identity)
        env = CARLPendulum(contexts=ContextTable(names, local), num_envs=sh.count, device=device,
                           context_selector=sel, lane_offset=sh.offset, max_episode_steps=7)
        obs, rew, idx = run(env, acts[:, sh.slice].contiguous())
        assert torch.equal(obs, f_obs[:, sh.slice]) and torch.equal(rew, f_rew[:, sh.slice]), (r, sh)
        want = f_idx[sh.slice] - (sh.offset if identity else 0)  # sharded rows are numbered from the shard's first
        assert torch.equal(idx, want.to(idx.dtype)), (r, idx, want)


_RCCL_WORLD1 = r"""
import os, sys, time
sys.path.insert(0, os.environ["CARL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from carl_amd.engine import VecEngine
from carl_amd.distributed import all_gather_episode_stats, reduce_episode_summary
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # backend "nccl" IS librccl on ROCm
d = np.load(os.environ["CARL_CASE"])
N = int(d["acts"].shape[1])
eng = VecEngine(int(d["fam"]), d["table"], N, dev, selector=0, seed=13, ctx_idx0=np.arange(N), max_episode_steps=int(d["max_steps"]))
eng.reset()
eng.rollout(torch.as_tensor(d["acts"], device=dev))
torch.cuda.synchronize()
assert eng.last_return.is_cuda
equal = all_gather_episode_stats(eng)                       # all_gather_into_tensor of DEVICE tensors
padded = all_gather_episode_stats(eng, counts=[N], padded=True)  # the uneven-shard form: pad + all_gather(list)
summary = reduce_episode_summary(eng)                        # all_reduce of four device scalars
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    all_gather_episode_stats(eng)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l or "libnccl" in l]
np.savez(os.environ["CARL_OUT"], ms=ms, rccl_loaded=len(loaded) > 0, world=dist.get_world_size(),
         backend=dist.get_backend(), summary_mean=summary["mean_return"], episodes=summary["episodes"],
         **{"eq_" + k: v.cpu().numpy() for k, v in equal.items()}, **{"pad_" + k: v.cpu().numpy() for k, v in padded.items()},
         **{"ref_" + k: getattr(eng, k).cpu().numpy() for k in ("last_return", "last_length", "episodes_done")})
dist.barrier()
dist.destroy_process_group()
"""


def test_rccl_collectives_run_on_this_gpu_with_a_world_of_one(device, tmp_path):
    """RCCL had never executed on hardware (two ranks on one device are refused).  A ONE-rank
    `nccl` process group on the one GPU: librccl is loaded, a communicator is built, and the engine's DEVICE tensors
    go through the reporting collectives -- the equal-count `all_gather_into_tensor` path, the padded list path of
    uneven shards, and the 4-scalar all-reduce.  With one rank every collective is the identity, so the results
    must equal the engine's own vectors bit for bit.  (Own process: the group must not leak into pytest's.)"""
    fam, N, T, max_steps = O.CARTPOLE, 8192, 64, 17
    rng = np.random.default_rng(22)
    case, out = tmp_path / "case.npz", tmp_path / "out.npz"
    np.savez(case, fam=fam, table=random_table(fam, rng, N), acts=random_actions(fam, rng, (T, N)), max_steps=max_steps)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29500 + (os.getpid() % 2000) + 11), CARL_ROOT=ROOT, CARL_CASE=str(case), CARL_OUT=str(out),
               CARL_AMD_NO_BUILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_WORLD1], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    got = np.load(out)
    assert str(got["backend"]) == "nccl" and int(got["world"]) == 1 and bool(got["rccl_loaded"])
    for k in ("last_return", "last_length", "episodes_done"):
        np.testing.assert_array_equal(got["eq_" + k], got["ref_" + k])
        np.testing.assert_array_equal(got["pad_" + k], got["ref_" + k])
    fin = got["ref_episodes_done"] > 0
    assert fin.any() and abs(float(got["summary_mean"]) - float(got["ref_last_return"][fin].mean())) < 1e-4
    assert float(got["episodes"]) == float(got["ref_episodes_done"].sum())
    print(f"RCCL world-1 all-gather of 3 x {N} device values: {float(got['ms']):.3f} ms per call")


def test_free_running_half_batches_equal_the_one_stream_rollouts(device):
    """`MixedVecEngine.rollout(free_running=True)` + `join()`: one family's contexts as two half-batches whose launch
    trains run on their own streams and overlap across calls (the double-buffered collector; bench.py's
    `free_running_two_streams` record).  Three consecutive free-running rollouts, one join: every output of every
    launch and the final engine state equal the same launches enqueued back to back on one stream."""
    from carl_amd.context.selection import StaticSelector
    from carl_amd.envs import CARLBraxAnt
    from carl_amd.mixed import MixedVecEngine

    n, T, R = 2048, 6, 3

    def mk():
        return [CARLBraxAnt(batch_size=n, device=device, context_selector=StaticSelector, seed=4, lane_offset=k * n,
                            autotune=False) for k in range(2)]
    a, b = mk(), mk()
    ma, mb = MixedVecEngine([p.env for p in a], ["ant", "ant"]), MixedVecEngine([p.env for p in b], ["ant", "ant"])
    for m_ in (ma, mb):
        m_.seed(4)
        m_.reset()
    g = torch.Generator(device=device).manual_seed(1)
    acts = [[torch.rand((T, n, 8), generator=g, device=device) * 2 - 1 for _ in range(2)] for _ in range(R)]
    outs_a = [ma.alloc_rollout(T) for _ in range(R)]
    outs_b = [mb.alloc_rollout(T) for _ in range(R)]
    for r in range(R):
        ma.rollout(acts[r], outs_a[r], free_running=True)   # no join between the calls
        mb.rollout(acts[r], outs_b[r])
    ma.join()
    torch.cuda.synchronize()
    for r in range(R):
        for k in range(2):
            for name in ("obs", "reward", "terminated", "truncated"):
                assert torch.equal(outs_a[r][k][name], outs_b[r][k][name]), (r, k, name)
    for k in range(2):
        assert torch.equal(ma.parts[k].state, mb.parts[k].state)
    assert torch.equal(ma.ep_return, mb.ep_return) and torch.equal(ma.episodes_done, mb.episodes_done)


def test_free_running_with_temporaries_and_auto_join(device):
    """a collector that builds a NEW action tensor each iteration and lets the engine allocate the
    outputs (``outs=None``) used to race with the caching allocator -- actions were allocated on the caller's stream and
    read on the parts' streams, outputs the other way round, neither recorded on the other stream.  Now both are
    (`record_stream`), and a `step` / `reset` / plain `rollout` issued while free-running launches are in flight joins
    first.  Many iterations with dropped temporaries + allocator churn in between == the one-stream engine."""
    from carl_amd.mixed import MixedVecEngine

    n, T, R = 4096, 16, 12
    fams, rng, mk = _pair(device, n, max_episode_steps=9)
    sep, parts = mk(), mk()
    mixed = MixedVecEngine(parts)
    for e in sep:
        e.reset()
    mixed.reset()
    g = torch.Generator(device=device).manual_seed(3)
    kept = []
    for r in range(R):
        acts = [torch.randint(0, 3, (T, n), generator=g, device=device, dtype=torch.int32) for _ in fams]  # temporaries
        outs = mixed.rollout(acts, None, free_running=True)
        ref = [e.rollout(a) for e, a in zip(sep, acts)]
        kept.append((outs, ref))
        del acts
        junk = [torch.empty(T * n, device=device, dtype=torch.int32).fill_(7) for _ in range(4)]  # allocator churn on the caller's stream
        del junk
    assert mixed._in_flight
    # a per-call step while launches are in flight: joins by itself, then equals the separate engines
    a1 = [torch.randint(0, 3, (n,), generator=g, device=device, dtype=torch.int32) for _ in fams]
    obs, rew, term, trunc = mixed.step(a1)
    assert not mixed._in_flight
    for k, e in enumerate(sep):
        o, r_, te, tr = e.step(a1[k])
        assert torch.equal(obs[k], o) and torch.equal(rew[mixed.part_slice(k)], r_)
    torch.cuda.synchronize()
    for outs, ref in kept:
        for k in range(2):
            for name in ("obs", "reward", "terminated", "truncated"):
                assert torch.equal(outs[k][name], ref[k][name]), (k, name)
    for k in range(2):
        for name in _BOOKKEEPING:
            assert torch.equal(getattr(parts[k], name), getattr(sep[k], name)), (k, name)


def test_small_brax_parts_side_by_side_equal_back_to_back(device):
    """The 8-GPU shard of BASELINE config 5 (Halfcheetah x 4 096 + Humanoid x 4 096 per GPU): `MixedVecEngine.rollout`
    launches such small Brax parts side by side on the parts' streams (fork / join inside the call) -- same bytes as
    back to back (`overlap=False`), and one stream-ordered operation for the caller."""
    from carl_amd.context.selection import StaticSelector
    from carl_amd.envs import CARLBraxHalfcheetah, CARLBraxHumanoid
    from carl_amd.mixed import MixedVecEngine

    n, T = 1024, 5

    def mk():
        return [cls(batch_size=n, device=device, context_selector=StaticSelector, seed=2, lane_offset=k * n, autotune=False)
                for k, cls in enumerate((CARLBraxHalfcheetah, CARLBraxHumanoid))]
    a, b = mk(), mk()
    ma, mb = MixedVecEngine([p.env for p in a]), MixedVecEngine([p.env for p in b])
    assert ma._small_brax_parts()
    for m_ in (ma, mb):
        m_.seed(2)
        m_.reset()
    g = torch.Generator(device=device).manual_seed(5)
    acts = [torch.rand((T, n, p.env.sys.n_act), generator=g, device=device) * 0.8 - 0.4 for p in a]
    o1 = ma.rollout(acts)                  # side by side (automatic)
    o2 = mb.rollout(acts, overlap=False)   # back to back
    after = o1[0]["reward"].sum() + o1[1]["reward"].sum()  # consumed on the caller's stream right away: must be ordered after both
    torch.cuda.synchronize()
    for k in range(2):
        for name in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(o1[k][name], o2[k][name]), (k, name)
        assert torch.equal(ma.parts[k].state, mb.parts[k].state)
    assert float(after) == float(o2[0]["reward"].sum() + o2[1]["reward"].sum())


def test_mixed_batch_with_uint8_actions_takes_separate_launches_same_results(device):
    """uint8 actions (ABI 7) are read by the single-family lean rollout; the heterogeneous pair kernel reads int32 --
    a mixed batch fed uint8 launches its parts one after the other, with the transitions of the int32 pair launch."""
    from carl_amd.engine import VecEngine
    from carl_amd.mixed import MixedVecEngine

    dev = device
    rng = np.random.default_rng(8)
    n, T = 1024, 19

    def build():
        parts = []
        for fam in (O.ACROBOT, O.MOUNTAINCAR):
            t = np.tile(O.default_row(fam), (n, 1))
            parts.append(VecEngine(fam, t, n, dev, selector=O.SEL_STATIC, seed=2, ctx_idx0=np.arange(n)))
        m = MixedVecEngine(parts)
        m.reset()
        return m

    m8, m32 = build(), build()
    a32 = [torch.as_tensor(rng.integers(0, 3, (T, n)).astype(np.int32), device=dev) for _ in range(2)]
    o8 = m8.rollout([a.to(torch.uint8) for a in a32])
    o32 = m32.rollout(a32)
    assert m8.pair_launches == 0 and m32.pair_launches == 1
    for p8, p32 in zip(o8, o32):
        for k in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(p8[k], p32[k]), k
