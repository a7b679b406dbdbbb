#!/usr/bin/env python
"""Headline benchmark: env-steps/sec at 65 536 parallel contexts per MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): CARLPendulum, 65 536 contexts per GPU sampled over
the features `g ~ U(1,20)` and `l ~ U(0.5,2)` (CARL's feature literally
called "gravity" is inert, Quirk P1), lane i <-> context i (StaticSelector), auto-reset
on, synthetic actions U(-2,2) resident in HBM.  One "step" = one env step of every lane
(65 536 env-steps per GPU).  The timed region runs the K steps through the engine's
fused entry point `carl_rollout` in launches of `--chunk` steps; every step writes its
complete transition (obs, reward, terminated, truncated) to HBM -- nothing is skipped.
The per-call path (`carl_step`, one launch per step, eager and hipGraph-replayed) is
measured right after and reported under "per_call" in the same JSON line.

Multi-GPU: weak scaling, lanes sharded by contiguous global-id ranges, no data-path
collective; one RCCL all-gather of the per-lane episodic returns after the timed region
(the reporting collective of SURVEY.md 8e), timed separately.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s achievable

# algorithmic bytes per env-step
# (a) SURVEY.md 8(d), per-call model (state/ctx/elapsed re-read every step)
BYTES_8D = {"pendulum": 66, "cartpole": 90, "acrobot": 110, "mountaincar": 74, "mountaincar_cont": 70,
            "ant": 1110, "halfcheetah": 4 * 6 + 4 * 17 + 6 + 2 * 13 * 7 * 4, "humanoid": 4 * 17 + 4 * 244 + 6 + 2 * 13 * 11 * 4}
# (b) fused rollout: per step only action in + transition out must cross HBM; state,
#     context params and counters cross once per launch (DESIGN.md "Kernels")
IO_PER_STEP = {"pendulum": 4 + 12 + 4 + 2, "cartpole": 4 + 16 + 4 + 2, "acrobot": 4 + 24 + 4 + 2,
               "mountaincar": 4 + 8 + 4 + 2, "mountaincar_cont": 4 + 8 + 4 + 2,
               "ant": 4 * 8 + 4 * 27 + 4 + 2, "halfcheetah": 4 * 6 + 4 * 17 + 4 + 2,
               "humanoid": 4 * 17 + 4 * 244 + 4 + 2}
PER_LAUNCH = {"pendulum": 8 + 4 + 16 + 4 + 4 + 8 + 4 + 4, "cartpole": 16 + 4 + 20 + 4 + 4 + 16 + 4 + 4,
              "acrobot": 16 + 4 + 36 + 4 + 4 + 4 + 16 + 4 + 4, "mountaincar": 8 + 4 + 28 + 4 + 4 + 8 + 4 + 4,
              "mountaincar_cont": 8 + 4 + 24 + 4 + 4 + 8 + 4 + 4,
              "ant": 2 * 13 * 9 * 4 + 4 + 5 * 4 + 4 + 4 + 4 + 4,
              "halfcheetah": 2 * 13 * 7 * 4 + 4 + 12 * 4 + 4 + 4 + 4 + 4,
              "humanoid": 2 * 13 * 11 * 4 + 4 + 15 * 4 + 4 + 4 + 4 + 4}


BRAX_ENVS = ("ant", "halfcheetah", "humanoid")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50000)
    p.add_argument("--warmup", type=int, default=5000)
    p.add_argument("--env", default="pendulum", choices=list(BYTES_8D))
    p.add_argument("--lanes", type=int, default=65536, help="lanes (= contexts) per GPU")
    p.add_argument("--chunk", type=int, default=250, help="env steps per fused launch")
    p.add_argument("--strong", action="store_true",
                   help="strong scaling: --lanes is the TOTAL number of contexts, split over the GPUs "
                        "(default: weak scaling, --lanes per GPU)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-per-call", action="store_true")
    p.add_argument("--cpu-envs-per-core", type=int, default=256)
    p.add_argument("--cpu-steps-per-env", type=int, default=1000)
    return p.parse_args()


def make_env(args, rank, world, device):
    import numpy as np

    from carl_amd.context.context_space import UniformFloatContextFeature as U
    from carl_amd.context.sampler import ContextSampler
    from carl_amd.context.selection import StaticSelector
    from carl_amd import envs as E

    cls = {"pendulum": E.CARLPendulum, "cartpole": E.CARLCartPole, "acrobot": E.CARLAcrobot,
           "mountaincar": E.CARLMountainCar, "mountaincar_cont": E.CARLMountainCarContinuous,
           "ant": E.CARLBraxAnt, "halfcheetah": E.CARLBraxHalfcheetah, "humanoid": E.CARLBraxHumanoid}[args.env]
    dists = {
        "pendulum": [U("g", 1, 20), U("l", 0.5, 2.0)],
        "cartpole": [U("gravity", 5, 15), U("length", 0.3, 1.0), U("masspole", 0.05, 0.3)],
        "acrobot": [U("LINK_LENGTH_1", 0.5, 2), U("LINK_MASS_1", 0.5, 2), U("LINK_MASS_2", 0.5, 2),
                    U("LINK_COM_POS_1", 0.3, 0.7), U("LINK_COM_POS_2", 0.3, 0.7)],
        "mountaincar": [U("force", 5e-4, 2e-3), U("gravity", 1.5e-3, 3.5e-3), U("goal_position", 0.3, 0.55)],
        "mountaincar_cont": [U("power", 5e-4, 3e-3), U("goal_position", 0.3, 0.55)],
        # BASELINE config 4
        "ant": [U("mass_torso", 5, 15), U("gravity", -15, -5), U("friction", 0.3, 1.5)],
        # BASELINE config 5
        "halfcheetah": [U("joint_stiffness", 0.5, 2.0), U("gravity", -15, -5), U("mass_torso", 5, 15)],
        "humanoid": [U("mass_torso", 5, 15), U("gravity", -15, -5), U("friction", 0.3, 1.5)],
    }[args.env]
    n = args.lanes // world if args.strong else args.lanes
    # one global context set (seed 0), each rank uploads only its lanes' rows
    table = ContextSampler(dists, cls.get_context_space(), seed=0).sample_context_table(n * world)
    from carl_amd.context.table import ContextTable

    local = ContextTable(table.names, table.values_2d[rank * n:(rank + 1) * n])
    size_kw = {"batch_size": n} if args.env in BRAX_ENVS else {"num_envs": n}
    env = cls(contexts=local, device=device, context_selector=StaticSelector, seed=0,
              lane_offset=rank * n, fin_capacity=0, **size_kw)
    return env, table


def make_actions(args, env, T, device, rank):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(1 + rank)
    info = env.env.info
    if info.action_is_discrete:
        return torch.randint(0, info.n_actions, (T, env.num_envs), generator=g, device=device, dtype=torch.int32)
    lo, hi = float(info.action_low), float(info.action_high)
    shape = (T, env.num_envs) if info.action_dim == 1 else (T, env.num_envs, int(info.action_dim))
    return torch.rand(shape, generator=g, device=device, dtype=torch.float32) * (hi - lo) + lo


def _cpu_worker(job):
    family, rows, names, steps = job
    from oracle import ref_style as R

    contexts = {i: dict(zip(names, r)) for i, r in enumerate(rows)}
    return R.time_loop(family, contexts, steps)


def _cpu_worker_brax(job):
    env, names, rows, steps = job
    import numpy as np

    from carl_amd.envs.brax.models import SYSTEMS
    from oracle import brax as B
    from oracle import oracle as O

    sys_t = SYSTEMS[env](names)
    rows = np.asarray(rows, dtype=np.float64)
    n = len(rows)
    eng = B.Engine(sys_t, rows, n, selector=O.SEL_STATIC, ctx_idx0=np.arange(n), seed=0)
    eng.reset()
    rng = np.random.default_rng(0)
    a = rng.uniform(-0.4, 0.4, (n, sys_t.n_act)).astype(np.float32)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(a)
    return n * steps, time.perf_counter() - t0


def cpu_baseline_brax(args, table):
    """Brax families: the fp64 C restatement of the spring pipeline (oracle/brax_spring.c), one
    process per host core, on a bounded sample of the same context set.  kind = "port" (brax /
    jax are not installable here)."""
    import multiprocessing as mp

    cores = os.cpu_count() or 1
    per, steps = 16, 200
    rows = table.values_2d
    names = list(table.names)
    jobs = [(args.env, names, rows[(c * per) % len(rows):(c * per) % len(rows) + per].tolist(), steps)
            for c in range(cores)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker_brax, [(args.env, names, j[2][:2], 2) for j in jobs])  # warm the workers
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker_brax, jobs)
        wall = time.perf_counter() - t0
    return {
        "value": sum(r[0] for r in res) / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"fp64 C oracle of the spring pipeline (oracle/brax_spring.c), {cores} processes x {per} contexts "
                  f"x {steps} env steps of the same {args.env} context set, auto-reset on",
        "single_core_value": res[0][0] / res[0][1],
    }


def cpu_baseline(args, table):
    """The reference-style scalar Python loop (oracle/ref_style.py: CARL wrapper ->
    TimeLimit -> env object, dict obs rebuilt per step) on all host cores, on a bounded
    sample of the same contexts.  kind = "port": the reference itself cannot run here
    (gymnasium is not installed)."""
    if args.env in BRAX_ENVS:
        return cpu_baseline_brax(args, table)
    import multiprocessing as mp

    import numpy as np

    from oracle import oracle as O

    fam = O.FAMILY_NAMES.index(args.env)
    cores = os.cpu_count() or 1
    per = args.cpu_envs_per_core
    F = len(O.FEATURES[fam])
    names = O.feature_names(fam)
    rows = table.values_2d[:, :F]
    jobs = [(fam, rows[(c * per) % len(rows):(c * per) % len(rows) + per].tolist(), names, args.cpu_steps_per_env)
            for c in range(cores)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(fam, j[1][:2], names, 10) for j in jobs])  # warm the workers (imports)
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    single = res[0][0] / res[0][1]
    # stronger CPU line: the vectorised C oracle (float64), one thread
    eng = O.Engine(fam, rows[: args.lanes].astype(np.float32).astype(np.float64), min(args.lanes, len(rows)),
                   selector=O.SEL_STATIC, precision="f64")
    eng.reset()
    a = np.zeros(eng.n, dtype=np.float32 if fam in O.CONTINUOUS else np.int32)
    t1 = time.perf_counter()
    k = 0
    while time.perf_counter() - t1 < 2.0:
        eng.step(a)
        k += 1
    c_rate = k * eng.n / (time.perf_counter() - t1)
    return {
        "value": total / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"reference-style scalar Python loop (oracle/ref_style.py), {cores} processes x {per} contexts "
                  f"x {args.cpu_steps_per_env} steps of the same {args.env} context set, auto-reset on",
        "single_core_value": single,
        "c_oracle_f64_1thread_value": c_rate,
    }


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def barrier():
        if dist is not None:
            dist.barrier()

    env, table = make_env(args, rank, world, device)
    eng = env.env
    n = env.num_envs
    K, W = args.steps, args.warmup
    T = max(1, min(args.chunk, K))
    chunks = [T] * (K // T) + ([K % T] if K % T else [])
    actions = make_actions(args, env, T, device, rank)
    out = eng.alloc_rollout(T)

    env.reset(seed=0)
    if args.env in BRAX_ENVS:  # launch shape chosen by timing on this batch (results do not depend on it)
        eng.autotune()
    done_w = 0
    while done_w < W:
        t = min(T, W - done_w)
        eng.rollout(actions[:t], out)
        done_w += t
    torch.cuda.synchronize()

    # ---- timed region: exactly K env steps of every lane -------------------------
    # one HIP-event pair around the whole launch train, on the stream the kernels run on
    # (torch's current stream): launches are back-to-back, so (elapsed / launches) is the
    # kernel's average duration plus the ~1-2 us kernel boundary
    e_first, e_last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e_first.record()
    for t in chunks:
        eng.rollout(actions[:t], out)
    e_last.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # average duration of a full-length launch (a shorter tail launch is pro-rated)
    avg_launch_s = e_first.elapsed_time(e_last) * 1e-3 * T / K
    bytes_per_step = IO_PER_STEP[args.env] + PER_LAUNCH[args.env] / T
    achieved = bytes_per_step * n * T / avg_launch_s / 1e9
    kernel_name = ("brax_kernel<1>" if args.env in BRAX_ENVS else
                   "rollout_staged_kernel" if n % 16 == 0 else "rollout_kernel")
    roofline = {
        "bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        "bytes_per_unit": bytes_per_step, "bytes_per_unit_survey_8d": BYTES_8D[args.env],
        "achieved_with_survey_8d_bytes": BYTES_8D[args.env] * n * T / avg_launch_s / 1e9,
        "avg_launch_ms": avg_launch_s * 1e3, "units_per_launch": n * T,
    }
    # HBM traffic per launch from the PMC passes of tools/profile_gpu.sh (FETCH_SIZE x 2
    # per the gfx950 correction + WRITE_SIZE), recorded under profiles/; a bench run cannot
    # collect counters itself, so this is the committed measurement for this exact workload
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f).get(f"{args.env}:{n}:{T}")
        if rec:
            roofline["traffic"] = rec["hbm_bytes_per_launch"]
            roofline["traffic_source"] = rec["source"]
    except OSError:
        pass

    # ---- reporting collective: episodic returns all-gathered over RCCL ------------
    gather_ms = None
    if dist is not None:
        from carl_amd.distributed import all_gather_episode_stats

        torch.cuda.synchronize()
        g0 = time.perf_counter()
        try:
            stats = all_gather_episode_stats(eng)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            mean_return = float(stats["last_return"].mean())
        except Exception as e:  # reporting collective only: never lose the measurement over it
            print(f"[bench] episodic-return all-gather failed: {e!r}", file=sys.stderr)
            mean_return = float(eng.last_return.mean())
    else:
        mean_return = float(eng.last_return.mean())

    # ---- per-call path (one launch per env step) -----------------------------------
    per_call = None
    if not args.no_per_call:
        Kc = min(K, 1000)
        a1 = actions[0].contiguous()
        for _ in range(50):
            eng.step(a1)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(Kc):
            eng.step(a1)
        torch.cuda.synchronize()
        eager = time.perf_counter() - t0
        per_call = {"eager_value": n * world * Kc / eager, "eager_ms_per_step": eager / Kc * 1e3,
                    "graph_value": None, "graph_ms_per_step": None, "roofline": None}
        # hipGraph replay of 100 step launches.  Skipped under multi-process RCCL (the NCCL
        # watchdog thread's event queries can invalidate a capture) and never fatal.
        if world == 1:
            try:
                graph = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream(device=device)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    eng.step(a1)
                    with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
                        for _ in range(100):
                            eng.step(a1)
                torch.cuda.current_stream().wait_stream(s)
                graph.replay()
                torch.cuda.synchronize()
                g0e, g1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = max(1, Kc // 100)
                t0 = time.perf_counter()
                g0e.record()
                for _ in range(reps):
                    graph.replay()
                g1e.record()
                torch.cuda.synchronize()
                gwall = time.perf_counter() - t0
                per_step_s = g0e.elapsed_time(g1e) * 1e-3 / (reps * 100)
                b8d = BYTES_8D[args.env] + 8  # + running-return read/write the engine adds
                per_call.update({
                    "graph_value": n * world * reps * 100 / gwall, "graph_ms_per_step": per_step_s * 1e3,
                    "roofline": {"bound": "hbm", "kernel": "step_kernel", "bytes_per_unit": b8d,
                                 "achieved": b8d * n / per_step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": b8d * n / per_step_s / 1e9 / HBM_PEAK_GBS,
                                 "note": "duration = graph-replayed launch-to-launch period (includes the "
                                         "~1.5 us kernel boundary)"},
                })
            except Exception as e:  # graph capture is an optimisation of the measurement, not the product
                per_call["graph_error"] = repr(e)[:200]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, table)

    if rank == 0:
        line = {
            "metric": "env-steps/sec (whole node) at 65k parallel contexts per GPU",
            "value": n * world * K / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"CARL{args.env} x {n} contexts/GPU, StaticSelector "
                                   f"lane<->context, auto-reset, fused carl_rollout in launches of {T} steps, "
                                   "full transition written per step",
                       "lanes_per_gpu": n, "total_lanes": n * world, "chunk": T, "parallelism": f"lane-shard x{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "per_call": per_call,
            "mean_last_episode_return": mean_return, "return_allgather_ms": gather_ms,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
