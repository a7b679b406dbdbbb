#!/usr/bin/env python
"""Headline benchmark: env-steps/sec (whole node) at 65 536 parallel contexts (BASELINE.json's metric).

    python bench.py --gpus 1 --steps 20 --warmup 5           (what the driver runs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload: north_star's target -- CARLCartPole with 65 536 sampled contexts (gravity, length, masspole;
carl/envs/gymnasium/classic_control/carl_cartpole.py:11-66), lane i <-> context i (StaticSelector), auto-reset on,
synthetic actions resident in HBM.  BASELINE.json configs[1] (CARLPendulum x 65 536 over g / l) and configs 3-5
follow under `also`.  With N GPUs the 65 536 contexts are SPLIT over the node (strong scaling -- the letter of "at 65k
parallel contexts (whole node)"): lanes are independent units sharded over the ranks by contiguous global-id
ranges with no data-path collective, `value` is the whole-node aggregate; the same JSON line carries the weak-scaling
run (every GPU steps its own 65 536 contexts -- the operating point the engine is built for) under `weak`.  `--weak`
swaps the two.

One "step" of this benchmark = ONE PASS of the hot path over one batch of synthetic input:
one fused `carl_rollout` launch that advances every lane by `--chunk` (1 000: SURVEY.md 8d's K) env steps and
writes every step's complete transition (obs, reward, terminated, truncated) to HBM --
nothing is skipped.  `--steps K` times exactly K such launches after `--warmup W` untimed
ones; `value` = lanes x chunk x K / elapsed (env-steps/s), `ms_per_step` = per launch,
`config.env_steps_per_step` = lanes x chunk.  Launches rotate through `--buffer-sets` (2)
action / output buffer sets (CartPole: 2 x 1.7 GB >> the 256 MB Infinity Cache), so the
stream is an HBM stream from the first timed launch on.

Also in the same JSON line:
  roofline      the fused kernel against the HBM peak, from HIP events on the launch stream
  cpu_baseline  the reference-style scalar Python loop on the host cores (kind "port":
                gymnasium / brax are not installable here, the reference cannot be timed)
  per_call      the one-launch-per-env-step path (`carl_step`, eager and hipGraph-replayed)
  also          the same launch train for north_star's target env (CARLCartPole x 65 536) and
                BASELINE configs 3-5 (Acrobot+MountainCar mixed batch, Ant, Halfcheetah+Humanoid)

  sustained     the same launch train for >= 0.3 s (the K-launch region of the driver's command is ~1.5 ms: burst
                clocks; sustained runs clock ~8 % lower)
  weak          (N > 1) 65 536 contexts PER GPU, same launch train -- `value` itself is the strong-scaling run
                (BASELINE's "at 65k parallel contexts (whole node)": 65 536 / N contexts per GPU); `--weak` swaps them
  repetitions   the K-launch timed region is run `--reps` (5) times; `value` / `ms_per_step` are the MEDIAN region
                (each region is exactly K launches between barrier + synchronize), all regions are listed
  clocks        sclk / mclk of this GPU sampled from sysfs while the sustained train runs (a 12 % swing between two
                runs of one binary can then be told from a regression)
  shard8        (N = 1) the 8-GPU operating point of BASELINE's configs measured on this ONE GPU: CARLCartPole /
                CARLPendulum x 8 192, CARLBraxAnt x 4 096, CARLBraxHalfcheetah x 4 096 + CARLBraxHumanoid x 4 096,
                each with predicted_node_value = 8 x and the implied strong-scaling efficiency

Multi-GPU: lanes sharded by contiguous global-id ranges, no data-path collective; one RCCL all-gather of the
per-lane episodic returns after the timed region (the reporting collective of SURVEY.md 8e), timed separately
(`--rccl` builds the one-rank RCCL group on a single GPU too, so that librccl runs on the one-GPU boxes).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
BENCH_VERSION = 4       # r04: headline CartPole x 65 536 (as r03) in 1 000-step launches (SURVEY 8d's K; r03: 250 -> `also.cartpole_T250`);
                        # value = median of --reps regions; strong scaling is `value` for N > 1; shard8 / kernel-time / VALU
                        # records added.  r03 = 3, r02 = 2 (Pendulum headline)

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
              # Brax: the env's 20 L-float record (pose head + tail + velocities, include/carl_amd.h) in and out
              "ant": 2 * 20 * 9 * 4 + 4 + 5 * 4 + 4 + 4 + 4 + 4,
              "halfcheetah": 2 * 20 * 7 * 4 + 4 + 12 * 4 + 4 + 4 + 4 + 4,
              "humanoid": 2 * 20 * 11 * 4 + 4 + 15 * 4 + 4 + 4 + 4 + 4}

BRAX_ENVS = ("ant", "halfcheetah", "humanoid")
# env steps per fused launch: SURVEY.md 8(d)'s K = 1 000 measured steps for the classic families (actions pre-generated
# on the device as [K, N]) -- one launch carries all of them (r02 / r03 used 250: `also.cartpole_T250` keeps that figure
# comparable); Brax: 20 env steps (= 200-320 pipeline substeps) per launch
DEFAULT_CHUNK = {e: (20 if e in BRAX_ENVS else 1000) for e in BYTES_8D}
# the other BASELINE workloads, run after the headline one (same launch train, fewer words):
#   name -> (families, total lanes per family, "weak" = per GPU / "strong" = split over the GPUs)
#   mode "follow": like the headline (strong by default: the total is split over the GPUs; --weak: per GPU);
#   "strong": always split (BASELINE defines configs 4 / 5 over the node).  Fourth entry: env steps per launch (None = default)
ALSO = {
    "cartpole": (("cartpole",), 65536, "follow", None),                # north_star's target env (the default headline)
    "cartpole_T250": (("cartpole",), 65536, "follow", 250),           # r02 / r03's launch length (4 x more launch boundaries)
    "cartpole_u8": (("cartpole",), 65536, "follow", None),             # the same workload fed uint8 actions (ABI 7: one byte
                                                                       # per lane-step on the only per-step read stream)
    "pendulum": (("pendulum",), 65536, "follow", None),                # BASELINE config 2
    "pendulum_f16": (("pendulum",), 65536, "follow", None),            # ... fed float16 torques (ABI 7: two bytes per lane-step)
    "config3": (("acrobot", "mountaincar"), 65536, "follow", None),    # 131 072-context mixed batch
    "config4": (("ant",), 32768, "strong", None),                      # 32 768 contexts over the node
    "config5": (("halfcheetah", "humanoid"), 32768, "strong", None),   # 65 536 contexts over the node
    # the same two workloads in 100-step launches (round 6): a Brax launch pays its set-up (model table -> LDS, state in /
    # out) and ~one env step per fragment boundary ONCE, so the 20-step figure above carries 5-10 % of per-launch cost that a
    # collector with longer unrolls does not pay.  Reported BESIDE config4 / config5 (whose definition stays: comparable
    # with rounds 2-5), never instead of them.  Not in the default `--also` list (they measured: config 4 +-0 %, config 5
    # +2.8 % -- the 20-step definition hides no per-launch cost -- and cost the default run half a minute): ask with --also.
    "config4_T100": (("ant",), 32768, "strong", 100),
    "config5_T100": (("halfcheetah", "humanoid"), 32768, "strong", 100),
    # OPT-IN, labelled, never the headline and never config4 / config5 themselves (VERDICT r05 "Next" #7): the substeps'
    # pose algebra in float32 -- brax's own precision under JAX's default -- with the deviation it costs beside the rate
    # (CARL_FLAG_BRAX_FP32; the product path above forms pose differences in float64 to meet north_star's 1e-5)
    "config4_fp32": (("ant",), 32768, "strong", None),
    "config5_fp32": (("halfcheetah", "humanoid"), 32768, "strong", None),
}
# the 8-GPU operating point of the BASELINE configs, measured on ONE GPU (N = 1 runs only): what each GPU of a node
# holds when BASELINE's totals are split eight ways.  name -> (families, lanes per family per GPU, full-size record)
SHARD8 = {
    "cartpole_8192": (("cartpole",), 8192, "cartpole"),
    "pendulum_8192": (("pendulum",), 8192, "pendulum"),
    "config3_8192+8192": (("acrobot", "mountaincar"), 8192, "config3"),
    "config4_ant_4096": (("ant",), 4096, "config4"),
    "config5_halfcheetah_4096+humanoid_4096": (("halfcheetah", "humanoid"), 4096, "config5"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200, help="timed fused launches (each = --chunk env steps of every lane)")
    p.add_argument("--warmup", type=int, default=20, help="untimed launches before them")
    p.add_argument("--env", default="cartpole",
                   help="family, or a+b for a mixed batch (e.g. acrobot+mountaincar); one of " + ", ".join(BYTES_8D))
    p.add_argument("--lanes", type=int, default=65536, help="lanes (= contexts) per family per GPU")
    p.add_argument("--chunk", type=int, default=0, help="env steps per fused launch (default 1000; Brax 20)")
    p.add_argument("--buffer-sets", type=int, default=2, help="action/output buffer sets the launches rotate through")
    p.add_argument("--strong", action="store_true", help="(default since r04; kept for older command lines)")
    p.add_argument("--weak", action="store_true",
                   help="headline value = weak scaling (--lanes contexts PER GPU).  Default: strong -- --lanes is the TOTAL "
                        "number of contexts (BASELINE: 65 536 over the whole node), split over the GPUs; for N > 1 the other "
                        "mode is reported in the same line (under 'weak' / 'strong') either way")
    p.add_argument("--reps", type=int, default=5, help="repetitions of the K-launch timed region (value = the median region)")
    p.add_argument("--no-shard8", action="store_true")
    p.add_argument("--lanes-per-env", default="",
                   help="Brax launch shape pinned instead of autotuned, e.g. 'ant=9,halfcheetah=7,humanoid=11' (counter passes: "
                        "the autotune probes would otherwise be averaged into the per-kernel counters)")
    p.add_argument("--rccl", action="store_true",
                   help="single GPU: build a one-rank RCCL process group and run the reporting all-gather through it")
    p.add_argument("--sustained-seconds", type=float, default=0.3)
    p.add_argument("--also", default="cartpole_T250,cartpole_u8,pendulum,pendulum_f16,config3,config4,config5,config4_fp32,config5_fp32",
                   help="comma list of extra workloads reported under 'also' (" + ", ".join(ALSO) + "), or 'none'")
    p.add_argument("--narrow-actions", action="store_true",
                   help="feed the MAIN workload uint8 (discrete) / float16 (Box) actions (ABI 7); profiling runs of "
                        "also.cartpole_u8 / pendulum_f16 -- the default headline reads int32")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-per-call", action="store_true")
    p.add_argument("--cpu-envs-per-core", type=int, default=128)
    p.add_argument("--cpu-steps-per-env", type=int, default=1000)
    a = p.parse_args()
    a.strong = not a.weak
    Workload.pinned_lanes_per_env = {k: int(v) for k, v in (kv.split("=") for kv in a.lanes_per_env.split(",") if kv)}
    a.families = tuple(a.env.split("+"))
    for f in a.families:
        if f not in BYTES_8D:
            p.error(f"unknown env {f}")
    return a


def context_dists(env):
    from carl_amd.context.context_space import UniformFloatContextFeature as U

    return {
        "pendulum": [U("g", 1, 20), U("l", 0.5, 2.0)],
        "cartpole": [U("gravity", 5, 15), U("length", 0.3, 1.0), U("masspole", 0.05, 0.3)],
        "acrobot": [U("LINK_LENGTH_1", 0.5, 2), U("LINK_MASS_1", 0.5, 2), U("LINK_MASS_2", 0.5, 2),
                    U("LINK_COM_POS_1", 0.3, 0.7), U("LINK_COM_POS_2", 0.3, 0.7)],
        "mountaincar": [U("force", 5e-4, 2e-3), U("gravity", 1.5e-3, 3.5e-3), U("goal_position", 0.3, 0.55)],
        "mountaincar_cont": [U("power", 5e-4, 3e-3), U("goal_position", 0.3, 0.55)],
        # BASELINE config 4
        "ant": [U("mass_torso", 5, 15), U("gravity", -15, -5), U("friction", 0.3, 1.5)],
        # BASELINE config 5
        # (not mass_torso: below ~0.75 x its default the torso's effective mass makes the explicit spring
        #  integration of this model unstable -- CARLBraxEnv clamps or refuses such contexts)
        "halfcheetah": [U("joint_stiffness", 0.5, 2.0), U("gravity", -15, -5)],
        "humanoid": [U("joint_stiffness", 0.5, 2.0), U("gravity", -15, -5)],
    }[env]


def make_env(env, n, rank, world, device, lane_base=0, brax_fp32=False):
    """One family: `n` lanes on this rank = global lanes [lane_base + rank n, lane_base + (rank+1) n) of one
    global context set (seed 0); each rank uploads only its lanes' rows."""
    from carl_amd import envs as E
    from carl_amd.context.sampler import ContextSampler
    from carl_amd.context.selection import StaticSelector
    from carl_amd.context.table import ContextTable

    cls = {"pendulum": E.CARLPendulum, "cartpole": E.CARLCartPole, "acrobot": E.CARLAcrobot,
           "mountaincar": E.CARLMountainCar, "mountaincar_cont": E.CARLMountainCarContinuous,
           "ant": E.CARLBraxAnt, "halfcheetah": E.CARLBraxHalfcheetahStiffness,
           "humanoid": E.CARLBraxHumanoidStiffness}[env]  # config 5: the classes WITH the joint_stiffness feature
    table = ContextSampler(context_dists(env), cls.get_context_space(), seed=0).sample_context_table(n * world)
    local = ContextTable(table.names, table.values_2d[rank * n:(rank + 1) * n])
    size_kw = {"batch_size": n, "autotune": False} if env in BRAX_ENVS else {"num_envs": n}
    if env in BRAX_ENVS and brax_fp32:
        size_kw["substep_precision"] = "float32"
    carl_env = cls(contexts=local, device=device, context_selector=StaticSelector, seed=0,
                   lane_offset=lane_base + rank * n, context_offset=lane_base + rank * n, fin_capacity=0, **size_kw)
    return carl_env, table


def make_actions(eng, T, device, seed, narrow=False):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    info = eng.info
    if info.action_is_discrete:
        a = torch.randint(0, info.n_actions, (T, eng.n), generator=g, device=device, dtype=torch.int32)
        return a.to(torch.uint8) if narrow else a  # (the same action values either way)
    lo, hi = float(info.action_low), float(info.action_high)
    shape = (T, eng.n) if info.action_dim == 1 else (T, eng.n, int(info.action_dim))
    a = torch.rand(shape, generator=g, device=device, dtype=torch.float32) * (hi - lo) + lo
    return a.to(torch.float16) if narrow and info.action_dim == 1 else a  # (narrow format: classic Box families only)


class Workload:
    """One or several families on this rank, their rotating action/output buffer sets, and the launch train."""

    pinned_lanes_per_env: dict = {}  # --lanes-per-env

    def __init__(self, families, lanes_per_gpu, T, sets, rank, world, device, narrow_actions=False, brax_fp32=False):
        import torch

        from carl_amd.mixed import MixedVecEngine

        self.families, self.T, self.device, self.narrow_actions = tuple(families), T, device, narrow_actions
        self.brax_fp32 = brax_fp32
        self.envs, self.tables = [], []
        for k, f in enumerate(self.families):
            e, t = make_env(f, lanes_per_gpu, rank, world, device, lane_base=k * lanes_per_gpu * world, brax_fp32=brax_fp32)
            self.envs.append(e)
            self.tables.append(t)
        self.mixed = len(self.families) > 1
        self.eng = MixedVecEngine([e.env for e in self.envs], self.families) if self.mixed else self.envs[0].env
        parts = self.eng.parts if self.mixed else [self.eng]
        self.n = sum(p.n for p in parts)
        self.acts = [[make_actions(p, T, device, 1 + rank + 1000 * s + 100 * k, narrow=narrow_actions) for k, p in enumerate(parts)]
                     for s in range(sets)]
        self.outs = [[p.alloc_rollout(T) for p in parts] for _ in range(sets)]
        for e in self.envs:
            e.reset(seed=0)
        if any(f in BRAX_ENVS for f in self.families):  # launch shape chosen by timing on this batch
            if all(f in self.pinned_lanes_per_env for f in self.families if f in BRAX_ENVS):
                for f, p in zip(self.families, parts):
                    if f in BRAX_ENVS:
                        p.sys.lanes_per_env = self.pinned_lanes_per_env[f]
            else:
                self.eng.autotune(n_steps=T)             # (results do not depend on it; probed at the launch length:
                                                         #  the best width for 20-step launches is not the 2-step one)
        torch.cuda.synchronize()
        self._i = 0
        self.units_per_launch = self.n * T
        act_saved = {f: ((3 if p.info.action_is_discrete else 2 if p.info.action_dim == 1 else 0) if narrow_actions else 0)
                     for f, p in zip(self.families, parts)}
        self.bytes_per_launch = sum(((IO_PER_STEP[f] - act_saved[f]) * T + PER_LAUNCH[f]) * p.n
                                    for f, p in zip(self.families, parts))
        self.bytes_per_launch_8d = sum(BYTES_8D[f] * T * p.n for f, p in zip(self.families, parts))

    def launch(self):
        s = self._i % len(self.acts)
        self._i += 1
        if self.mixed:
            self.eng.rollout(self.acts[s], self.outs[s])
        else:
            self.eng.rollout(self.acts[s][0], self.outs[s][0])

    def train(self, K, W, barrier, collect=True, per_launch=False):
        """W untimed launches, then exactly K timed ones -> (wall seconds incl. the syncs, average launch
        seconds from one HIP-event pair on the launch stream).  per_launch: a HIP event between every two launches as
        well -> additionally the K launch-to-launch times in seconds (a region of its own: the contract's regions
        carry one event pair, nothing between their launches)."""
        import torch

        import gc

        # The timed region is ~1.5 ms of host time: a generation-2 garbage collection landing inside it (the context
        # sets of the workloads are 65 536-row Python-side objects) stalls the launch loop for tens of milliseconds --
        # seen as a wall-clock value 10 x below the HIP-event figure of the same launches.  Collect BEFORE the warm-up
        # launches and keep the collector off while the clock runs.  (Round 3 collected between the warm-up and the timed
        # launches: the first timed launch then took 135-145 us instead of 85 -- cold host caches after the collection --
        # 3 % of a 20-launch region: tools/diag_region.py.)
        if collect:
            gc.collect()
        gc.disable()
        for _ in range(W):
            self.launch()
        torch.cuda.synchronize()
        # one HIP-event pair around the whole launch train, on the stream the kernels run on (torch's
        # current stream): launches are back-to-back, so elapsed / K is the kernel's average duration
        # plus the ~1-2 us kernel boundary
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = []
        e0.record()
        for _ in range(K):
            self.launch()
            if per_launch:
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()
        e1.record()
        torch.cuda.synchronize()
        barrier()
        wall = time.perf_counter() - t0
        gc.enable()
        if per_launch:
            ev = [e0] + marks
            return wall, e0.elapsed_time(e1) * 1e-3 / K, [ev[i].elapsed_time(ev[i + 1]) * 1e-3 for i in range(K)]
        return wall, e0.elapsed_time(e1) * 1e-3 / K

    def train_free_running(self, K, W, barrier):
        """The same K launches per part, each part on its OWN stream, joined only at the end
        (`MixedVecEngine.rollout(free_running=True)` + `join()`): consecutive launches of different parts overlap (the
        tail of one fills the head of the other).  What a double-buffered collector does (half-batch A steps while the
        policy looks at half-batch B); NOT the one-stream launch train `value` is quoted on -- reported beside it for
        the Brax workloads, whose workgroups live for a whole launch and leave the last 'round' of SIMD slots half
        empty."""
        import torch

        def go(n):
            for _ in range(n):
                s = self._i % len(self.acts)
                self._i += 1
                self.eng.rollout(self.acts[s], self.outs[s], free_running=True)
            self.eng.join()

        import gc

        torch.cuda.synchronize()
        go(W)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        go(K)
        torch.cuda.synchronize()
        barrier()
        wall = time.perf_counter() - t0
        gc.enable()
        return wall

    def launch_shape(self):
        """Brax families: the autotuned lane-group width per part (a pure scheduling choice)"""
        parts = self.eng.parts if self.mixed else [self.eng]
        return {f: int(p.sys.lanes_per_env) for f, p in zip(self.families, parts) if hasattr(p, "sys")}

    def kernel_name(self):
        names = []
        for f in self.families:
            names.append("brax_kernel<1>" if f in BRAX_ENVS else "rollout_staged_kernel<%s>" % f)
        return " + ".join(names)

    def mean_last_return(self):
        return float(self.eng.last_return.mean())


class SplitWorkload(Workload):
    """ONE family's contexts as two independent engines of half the lanes each, stepped as one `MixedVecEngine`: the
    parts of a free-running, double-buffered launch train."""

    def __init__(self, family, lanes_each, T, sets, rank, world, device):
        super().__init__((family, family), lanes_each, T, sets, rank, world, device)


def traffic_record(key):
    """HBM traffic per launch from the PMC passes of tools/profile_gpu.sh (FETCH_SIZE x 2 per the gfx950
    correction + WRITE_SIZE), recorded under profiles/: a bench run cannot collect counters itself, so this is
    the committed measurement for this exact workload (key = env:lanes:chunk)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(key)
    except OSError:
        return None



def profile_record(name, key):
    """A committed measurement a bench run cannot take itself (counters / kernel timestamps need rocprofv3 around the
    process): profiles/<name>.json, keyed by workload (env:lanes:chunk)."""
    try:
        with open(os.path.join(ROOT, "profiles", name + ".json")) as f:
            return json.load(f).get(key)
    except OSError:
        return None


class ClockSampler:
    """sclk / mclk of THIS GPU from sysfs (pp_dpm_sclk / pp_dpm_mclk: the level marked '*'), polled by a thread while a
    launch train runs.  The card is found through the device's PCI bus id; if that fails the record says so."""

    def __init__(self, device):
        import glob

        import torch

        self.dir = None
        try:
            bus = "%02x:00.0" % torch.cuda.get_device_properties(device).pci_bus_id
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(d).endswith(bus) and os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                    self.dir = d
                    break
        except Exception:
            pass
        self.samples = {"sclk": [], "mclk": [], "fclk": [], "socclk": []}
        # hwmon of the same card: temperatures (edge / junction / mem) and package power.  r04: on ONE box inside ONE
        # gpurun call the driver's command ran at 274 us per launch first and at 304 us a minute of GPU work later, with
        # sclk and mclk unchanged -- the "fast / medium / slow boxes" of the pool are at least partly the state a box is in
        self.hwmon = self.discover_hwmon(self.dir)
        self.hw_samples = {k: [] for k in self.hwmon}
        self._stop = False
        self._thread = None

    @staticmethod
    def discover_hwmon(card_dir):
        """{record key: (file, scale)} of the card's hwmon: temp*_input (millidegrees; labelled edge / junction / mem) and
        package power (microwatts)"""
        import glob

        found = {}
        if card_dir is None:
            return found
        for h in sorted(glob.glob(os.path.join(card_dir, "hwmon", "hwmon*"))):
            for f in sorted(glob.glob(os.path.join(h, "temp*_input"))):
                try:
                    with open(f.replace("_input", "_label")) as lf:
                        label = lf.read().strip()
                except OSError:
                    label = os.path.basename(f).split("_")[0]
                found["temp_" + label + "_c"] = (f, 1e-3)
            for name in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(h, name)):
                    found["power_w"] = (os.path.join(h, name), 1e-6)
                    break
        return found

    def _read(self, which):
        try:
            with open(os.path.join(self.dir, "pp_dpm_" + which)) as f:
                for line in f:
                    if "*" in line:
                        return int("".join(c for c in line.split(":")[1] if c.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def __enter__(self):
        if self.dir is not None:
            import threading

            def poll():
                while not self._stop:
                    for k in self.samples:
                        v = self._read(k)
                        if v is not None:
                            self.samples[k].append(v)
                    for k, (path, scale) in self.hwmon.items():
                        try:
                            with open(path) as f:
                                self.hw_samples[k].append(int(f.read()) * scale)
                        except (OSError, ValueError):
                            pass
                    time.sleep(0.01)

            self._thread = threading.Thread(target=poll, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join()

    def record(self):
        if self.dir is None:
            return {"source": None, "note": "sysfs clock files of this GPU not found"}
        out = {"source": "sysfs pp_dpm_{sclk,mclk,fclk,socclk} and hwmon temperatures / power of this card, polled every "
                         "10 ms during the sustained launch train"}
        for k, v in self.samples.items():
            if v:
                v = sorted(v)
                out[k + "_mhz"] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1], "samples": len(v)}
        for k, v in self.hw_samples.items():
            if v:
                v = sorted(v)
                out[k] = {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)}
        return out


def timed_regions(wl, K, W, reps, barrier, max_over_ranks):
    """`reps` repetitions of the contract's timed region (W warm-up launches before the first; each region = exactly K
    launches between barrier + synchronize).  Returns the per-region (wall max-over-ranks, HIP-event launch period) and
    the index of the median region by wall clock."""
    regions = []
    for r in range(max(1, reps)):
        wall, period = wl.train(K, W if r == 0 else 0, barrier, collect=(r == 0))
        regions.append((max_over_ranks(wall), period))
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    return regions, order[(len(order) - 1) // 2]


# SQ_INSTS_VALU x cycles per wavefront-instruction / (SIMDs x clock): the time the vector ALUs of the chip need to ISSUE
# the launch's instruction stream -- the compute roofline of the Brax kernels (VERDICT r03 #4a).  2.98 cycles per fp32
# wavefront-instruction at >= 2 wavefronts per SIMD: profiles/r03_fp64_rate.txt (the float64 / packed share issues at
# 4.7: the floor below is therefore a LOWER bound of the issue time, the fraction an upper bound of the distance left).
VALU_CYCLES_PER_INST = 2.98
N_SIMDS = 1024
SCLK_HZ = 2.4e9


def roofline_valu_of(key, avg_launch_s):
    rec = profile_record("brax_valu", key)
    if not rec:
        return None
    floor_s = rec["insts_valu_per_launch"] * VALU_CYCLES_PER_INST / N_SIMDS / SCLK_HZ
    spec_s = rec["insts_valu_per_launch"] * 2.0 / N_SIMDS / SCLK_HZ  # the guide's 2-cycle wave64 issue (157.3 TF vector peak)
    return {"bound": "valu-issue", "insts_valu_per_launch": rec["insts_valu_per_launch"],
            "cycles_per_inst": VALU_CYCLES_PER_INST, "simds": N_SIMDS, "sclk_hz": SCLK_HZ,
            "issue_floor_ms": floor_s * 1e3, "kernel_ms": avg_launch_s * 1e3, "frac": floor_s / avg_launch_s,
            "frac_at_spec_2_cycles": spec_s / avg_launch_s,
            "source": rec.get("source"), "insts_salu_per_launch": rec.get("insts_salu_per_launch"),
            "insts_lds_per_launch": rec.get("insts_lds_per_launch")}


def roofline_of(wl, avg_launch_s):
    achieved = wl.bytes_per_launch / avg_launch_s / 1e9
    r = {
        "bound": "hbm", "kernel": wl.kernel_name(), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        "bytes_per_unit": wl.bytes_per_launch / wl.units_per_launch,
        "bytes_per_unit_survey_8d": wl.bytes_per_launch_8d / wl.units_per_launch,
        "achieved_with_survey_8d_bytes": wl.bytes_per_launch_8d / avg_launch_s / 1e9,
        "avg_launch_ms": avg_launch_s * 1e3, "units_per_launch": wl.units_per_launch,
        "algorithmic_bytes_per_launch": wl.bytes_per_launch,
    }
    part_n = wl.n // len(wl.families)
    key = (f"{'+'.join(wl.families)}{'_narrow' if getattr(wl, 'narrow_actions', False) else ''}"
           f"{'_fp32' if getattr(wl, 'brax_fp32', False) else ''}:{part_n}:{wl.T}")  # (committed records are per kernel variant)
    rec = traffic_record(key)
    if rec:
        r["traffic"] = rec["hbm_bytes_per_launch"]
        r["traffic_source"] = rec["source"]
    # `frac` = frac_period: bytes / the launch-to-launch PERIOD of the timed region (one HIP-event pair over its K
    # launches; what a caller gets).  frac_launch_median: the same bytes / the MEDIAN launch-to-launch interval of THIS
    # process (a HIP event between every two launches of one further region: main()) -- still a period (kernel + dispatch
    # gap; the kernel trace shows a 0.00 us median gap inside a train), so it is NOT called a kernel figure.
    # Every fraction at this level comes from this process on this box.  profile_reference: the committed rocprofv3
    # --kernel-trace --stats figure of the same workload (another run, possibly another box: a bench run cannot trace
    # itself), with that run's own fraction = `frac_kernel` -- the only figure of the line that carries "kernel" in its
    # name is the one from a kernel trace; for reading beside profiles/, never mixed into this line's own numbers.
    r["frac_period"] = r["frac"]
    r["frac_launch_median"] = None
    kt = profile_record("kernel_times", key)
    if kt:
        fk = wl.bytes_per_launch / (kt["kernel_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        r["profile_reference"] = {"kernel_avg_us_rocprofv3": kt["kernel_avg_us"], "frac_kernel": fk,
                                  "frac_of_that_run": fk, "source": kt["source"]}
    return r


def launch_stats(wl, roofline, K, barrier):
    """One further region of K launches with a HIP event between every two -> min / median / max launch time of this
    process and `frac_launch_median` from the median (VERDICT r04 #2b; named for what it is: ADVICE r05)."""
    _, _, per = wl.train(K, 0, barrier, collect=False, per_launch=True)
    per = sorted(per)
    med = per[(len(per) - 1) // 2]
    roofline["launch_us"] = {"n": len(per), "min": per[0] * 1e6, "median": med * 1e6, "max": per[-1] * 1e6,
                             "how": "HIP event between every two launches of one extra K-launch region, this process"}
    roofline["frac_launch_median"] = wl.bytes_per_launch / med / 1e9 / HBM_PEAK_GBS
    return roofline


def _cpu_worker(job):
    family, rows, names, steps = job
    from oracle import ref_style as R

    contexts = {i: dict(zip(names, r)) for i, r in enumerate(rows)}
    return R.time_loop(family, contexts, steps)


def _cpu_worker_brax(job):
    sys_t, rows, steps = job  # sys_t: oracle.brax.SysSnapshot (the worker imports neither the product package nor torch)
    import numpy as np

    from oracle import brax as B
    from oracle import oracle as O

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


_POOL = None


def _get_pool():
    """ONE process pool (one worker per host core, spawn) for every CPU baseline of the run: starting 256 interpreters costs
    seconds, and round 6 times a baseline beside each BASELINE config."""
    global _POOL
    if _POOL is None:
        import multiprocessing as mp

        _POOL = mp.get_context("spawn").Pool(os.cpu_count() or 1)
    return _POOL


def _close_pool():
    global _POOL
    if _POOL is not None:
        _POOL.terminate()
        _POOL.join()
        _POOL = None


def cpu_baseline_brax(env, table):
    """Brax families: the fp64 C restatement of the spring pipeline (oracle/brax_spring.c), one
    process per host core, on a bounded sample of the same context set.  kind = "port" (brax /
    jax are not installable here)."""
    import multiprocessing as mp

    cores = os.cpu_count() or 1
    per, steps = 16, 200
    rows = table.values_2d
    names = list(table.names)
    from carl_amd.envs.brax.models import SYSTEMS
    from oracle import brax as B

    snap = B.SysSnapshot(SYSTEMS[env](names))
    jobs = [(snap, rows[(c * per) % len(rows):(c * per) % len(rows) + per].tolist(), steps) for c in range(cores)]
    pool = _get_pool()
    pool.map_async(_cpu_worker_brax, [(snap, j[1][:2], 2) for j in jobs], chunksize=1).get(timeout=180)  # warm the workers
    t0 = time.perf_counter()
    res = pool.map_async(_cpu_worker_brax, jobs, chunksize=1).get(timeout=300)  # (bounded: a stuck worker must not hang the bench)
    wall = time.perf_counter() - t0
    return {
        "value": sum(r[0] for r in res) / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"fp64 C restatement of the spring pipeline (oracle/brax_spring.c), {cores} processes x {per} contexts "
                  f"x {steps} env steps of the same {env} context set, auto-reset on",
        "single_core_value": res[0][0] / res[0][1],
    }


def cpu_baseline(args, env, table, lanes):
    """The reference-style scalar Python loop (oracle/ref_style.py: CARL wrapper ->
    TimeLimit -> env object, dict obs rebuilt per step) on all host cores, on a bounded
    sample of the same contexts.  kind = "port": a RESTATEMENT of the reference path -- the
    reference itself cannot run here (gymnasium is not installed)."""
    if env in BRAX_ENVS:
        return cpu_baseline_brax(env, table)
    import multiprocessing as mp

    import numpy as np

    from oracle import oracle as O

    fam = O.FAMILY_NAMES.index(env)
    cores = os.cpu_count() or 1
    # a BOUNDED sample (about ten seconds per family on the reference-style Python loop): the headline family keeps the
    # command line's sample; the others are scaled by what their scalar step costs (Acrobot's RK4 in Python: ~15 x CartPole's)
    scale = {"cartpole": (1, 1), "pendulum": (2, 2), "acrobot": (8, 4), "mountaincar": (2, 1), "mountaincar_cont": (2, 1)}[env]
    head = env == args.families[0]
    per = args.cpu_envs_per_core if head else max(8, args.cpu_envs_per_core // scale[0])
    steps = args.cpu_steps_per_env if head else max(50, args.cpu_steps_per_env // scale[1])
    F = len(O.FEATURES[fam])
    names = O.feature_names(fam)
    rows = table.values_2d[:, :F]
    jobs = [(fam, rows[(c * per) % len(rows):(c * per) % len(rows) + per].tolist(), names, steps)
            for c in range(cores)]
    pool = _get_pool()
    pool.map_async(_cpu_worker, [(fam, j[1][:2], names, 10) for j in jobs], chunksize=1).get(timeout=180)  # warm the workers (imports)
    t0 = time.perf_counter()
    res = pool.map_async(_cpu_worker, jobs, chunksize=1).get(timeout=300)  # (bounded: a stuck worker must not hang the bench)
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    single = res[0][0] / res[0][1]
    # stronger CPU line: the vectorised C oracle (float64), one thread
    eng = O.Engine(fam, rows[:lanes].astype(np.float32).astype(np.float64), min(lanes, len(rows)),
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
        "sample": f"restatement of the reference's scalar Python step() loop (oracle/ref_style.py), {cores} processes x "
                  f"{per} contexts x {steps} steps of the same {env} context set, auto-reset on",
        "single_core_value": single,
        "c_oracle_f64_1thread_value": c_rate,
    }


def brax_fp32_deviation(env, device, n=4096, warm=8):
    """What CARL_FLAG_BRAX_FP32 costs in accuracy: ONE env step (n_frames substeps) from identical states through the float64
    and the float32 substeps of this library, `warm` random-policy steps away from reset; |d| / (1 + |x|) over every
    observation entry.  The float64 path is the one the parity tests hold within 1e-5 of the float64 restatement; the
    deviation from the restatement itself (same order) is in profiles/r06_brax_fp32_deviation.txt."""
    import torch

    e64, _ = make_env(env, n, 0, 1, device)
    e32, _ = make_env(env, n, 0, 1, device, brax_fp32=True)
    a, b = e64.env, e32.env
    e64.reset(seed=0)
    e32.reset(seed=0)
    g = torch.Generator(device=device).manual_seed(11)
    lo, hi = float(min(a.sys.act_lo[: a.sys.n_act])), float(max(a.sys.act_hi[: a.sys.n_act]))
    qs, worst = [], 0.0
    for t in range(warm + 4):
        act = torch.rand((n, a.sys.n_act), generator=g, device=device) * (hi - lo) + lo
        b._state_storage.copy_(a._state_storage)
        for k in ("elapsed", "ep_return", "episode", "n_calls", "ctx_idx"):
            getattr(b, k).copy_(getattr(a, k))
        o64 = a.step(act)[0].clone()
        o32 = b.step(act)[0]
        if t >= warm:
            d = ((o32 - o64).abs() / (1 + o64.abs())).flatten()
            qs.append(torch.quantile(d, torch.tensor([0.5, 0.99, 0.9999], device=device)).cpu())
            worst = max(worst, float(d.max()))
    q = torch.stack(qs).mean(0)
    return {"p50": float(q[0]), "p99": float(q[1]), "p99.99": float(q[2]), "max": worst, "envs": n, "steps_compared": 4,
            "what": "|obs_f32 - obs_f64| / (1 + |obs_f64|) after ONE env step from identical states (auto-reset on both)"}


def uneven_shard_record(env, total, rank, world, device, backend, barrier, gather_over_ranks):
    """`total` lanes (not a multiple of the world size) split by carl_amd.distributed.lane_shard: one 64-step rollout
    per rank on its shard, then the episodic-return all-gather with uneven counts.  A plumbing record (which kernel each
    shard took, that the gathered vector has one entry per global lane in lane order), not a throughput claim."""
    import numpy as np
    import torch

    from carl_amd import _lib
    from carl_amd import envs as E
    from carl_amd.context.sampler import ContextSampler
    from carl_amd.context.selection import StaticSelector
    from carl_amd.context.table import ContextTable
    from carl_amd.distributed import all_gather_episode_stats, lane_shard

    if env in BRAX_ENVS:
        return None
    cls = {"pendulum": E.CARLPendulum, "cartpole": E.CARLCartPole, "acrobot": E.CARLAcrobot,
           "mountaincar": E.CARLMountainCar, "mountaincar_cont": E.CARLMountainCarContinuous}[env]
    sh = lane_shard(total, rank, world)
    table = ContextSampler(context_dists(env), cls.get_context_space(), seed=0).sample_context_table(total)
    local = ContextTable(table.names, table.values_2d[sh.slice])
    e = cls(contexts=local, device=device, context_selector=StaticSelector, seed=0, lane_offset=sh.offset,
            context_offset=sh.offset, fin_capacity=0, num_envs=sh.count)
    e.reset(seed=0)
    eng = e.env
    T = 64
    out = eng.rollout(make_actions(eng, T, device, 7 + rank))
    torch.cuda.synchronize()
    counts = [lane_shard(total, r, world).count for r in range(world)]
    src = eng if backend == "nccl" else {k: getattr(eng, k).cpu() for k in ("last_return", "last_length", "episodes_done")}
    barrier()
    t0 = time.perf_counter()
    stats = all_gather_episode_stats(src, counts=counts)
    if backend == "nccl":
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    # lane order: every rank's own slice of the gathered vector is its local vector
    mine = stats["episodes_done"][sh.offset: sh.offset + sh.count].to(eng.episodes_done.device if backend == "nccl" else "cpu")
    ok = bool(torch.equal(mine, eng.episodes_done if backend == "nccl" else eng.episodes_done.cpu()))
    variants = gather_over_ranks(float(eng.rollout_variant()))
    return {"total_lanes": total, "counts": counts, "gathered": int(stats["last_return"].numel()),
            "lane_order_ok_on_rank0": ok, "allgather_ms": ms, "row_pitch_rank0": int(out["reward"].stride(0)),
            "rollout_variant_per_rank": [int(v) for v in variants], "staged": int(_lib.ROLLOUT_STAGED),
            "env_steps_per_rank": T}


# `also` records that carry their own CPU baseline: north_star's env and BASELINE configs 2 - 5
CPU_BESIDE = ("cartpole", "pendulum", "config3", "config4", "config5")


def cpu_baseline_beside(args, fams, tables, lanes):
    """`cpu_baseline` for one `also` workload: the single family's record, or -- a mixed batch -- one record per family
    and the rate at which the host cores would step the WHOLE batch (every family's lanes once per step: harmonic
    combination of the per-family rates, equal lanes per family).  A failure is recorded, never raised: the CPU figure is
    a reported baseline and must not take the GPU measurement with it."""
    recs = {}
    for f, tab in zip(fams, tables):
        try:
            recs[f] = cpu_baseline(args, f, tab, lanes)
        except Exception as e:
            _close_pool()  # (a stuck worker must not stall the next baseline too: the next one starts a fresh pool)
            recs[f] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "error": repr(e)[:200]}
    if len(fams) == 1:
        return recs[fams[0]]
    vals = [recs[f].get("value") for f in fams]
    combined = (len(vals) / sum(1.0 / v for v in vals)) if all(vals) else None
    return {"value": combined, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "per family below (bounded samples of the same context sets); value = the rate of stepping every "
                      "family's lanes once per step on all host cores = n / sum(1 / rate_f), equal lanes per family",
            "per_family": recs}


def per_call_record(eng, action, n_total, Kc, device, world, barrier):
    """The one-launch-per-env-step path a policy-in-the-loop caller uses: eager `step`, and the engine's own
    captured hipGraph (`capture_step`) replayed."""
    import torch

    for _ in range(50):
        eng.step(action)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(Kc):
        eng.step(action)
    torch.cuda.synchronize()
    eager = time.perf_counter() - t0
    rec = {"eager_value": n_total * Kc / eager, "eager_ms_per_step": eager / Kc * 1e3,
           "graph_value": None, "graph_ms_per_step": None, "roofline": None}
    return rec


def graph_per_call(rec, eng, action, env, n, Kc, device):
    """hipGraph replay of 100 step launches (single process only: under multi-process RCCL the NCCL watchdog
    thread's event queries can invalidate a capture)."""
    import torch

    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=device)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.step(action)
        with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
            for _ in range(100):
                eng.step(action)
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
    b8d = BYTES_8D[env] + 8  # + running-return read/write the engine adds
    rec.update({
        "graph_value": n * reps * 100 / gwall, "graph_ms_per_step": per_step_s * 1e3,
        "roofline": {"bound": "hbm", "kernel": "step_kernel", "bytes_per_unit": b8d,
                     "achieved": b8d * n / per_step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": b8d * n / per_step_s / 1e9 / HBM_PEAK_GBS,
                     "duration_us": per_step_s * 1e6,
                     "note": "duration = launch-to-launch period of back-to-back launches (hipGraph replay, HIP events): "
                             "what a caller pays per env step.  rocprofv3's per-dispatch duration of this kernel is "
                             "LONGER (3.7 us median): the profiler serialises dispatches and times each one alone "
                             "(ramp-up + one HBM round trip + release), phases that back-to-back launches overlap "
                             "-- profiles/r03_per_call_period.txt",
                     "isolated_dispatch_us_rocprofv3": 3.68,
                     "frac_isolated_dispatch": b8d * n / 3.68e-6 / 1e9 / HBM_PEAK_GBS},
    })


def dropin_per_call(device, n_big, cpu_single_core_value):
    """The drop-in boundary's own cost per env step: what a maintainer of the reference binds is not the
    engine but `Mi355xVecEnv` under the reference's `CARLEnv` (carl/envs/carl_env.py:321-342 ->
    carl/envs/gymnasium/carl_gymnasium_env.py:63-77), and the mirror class `carl_amd.envs.CARLCartPole`.
      scalar:  num_envs = 1, the reference's own calling convention -- Python action in, float32 ndarray / float / bool
               out, i.e. one launch AND one device-to-host read per env step;
      batched: num_envs = n_big, device tensors in and out, no host read inside the loop."""
    import numpy as np
    import torch

    from carl_amd.dropin import Mi355xVecEnv
    from carl_amd.envs import CARLCartPole

    rec = {"unit": "us per step() call", "family": "CartPole-v1",
           "cpu_reference_style_us_per_step": (1e6 / cpu_single_core_value) if cpu_single_core_value else None}
    env = Mi355xVecEnv("CartPole-v1", num_envs=1, device=device)
    env.reset(seed=0)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 2, 2200)
    for a in acts[:200]:
        _, _, term, trunc, _ = env.step(int(a))
        if term or trunc:
            env.reset()
    t0 = time.perf_counter()
    for a in acts[200:]:
        _, _, term, trunc, _ = env.step(int(a))
        if term or trunc:
            env.reset()
    rec["scalar_num_envs_1"] = (time.perf_counter() - t0) / 2000 * 1e6
    del env
    big = Mi355xVecEnv("CartPole-v1", num_envs=n_big, device=device)
    big.reset(seed=0)
    a = torch.randint(0, 2, (n_big,), device=device, dtype=torch.int32)
    for _ in range(100):
        big.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        big.step(a)
    torch.cuda.synchronize()
    rec[f"batched_num_envs_{n_big}"] = (time.perf_counter() - t0) / 1000 * 1e6
    del big
    mirror = CARLCartPole(num_envs=n_big, device=str(device), seed=0)
    mirror.reset(seed=0)
    for _ in range(100):
        mirror.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        mirror.step(a)
    torch.cuda.synchronize()
    rec[f"mirror_CARLCartPole_num_envs_{n_big}"] = (time.perf_counter() - t0) / 1000 * 1e6
    rec["note"] = ("scalar = one launch + one device-to-host read per env step (the reference's return types; pinned action "
                   "staging, one copy for the whole transition); cpu_reference_style_us_per_step is the loop it replaces -- the "
                   "shim pays off from num_envs > 1 (INTEGRATION.md)")
    return rec


class _stdout_to_stderr:
    """RCCL prints a version banner with C stdio on stdout when a communicator is created; the contract of this
    script is ONE JSON line on stdout.  File descriptor 1 points at stderr while the group is built (and the C
    buffers are flushed before it is restored)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes

        ctypes.CDLL(None).fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)


def launch_plan(gpus, env, n_devices):
    """What `bench.py --gpus N` does, given the environment it was started in (pure function: tests/test_bench_contract.py).
    ("run", None): this process is the rank the environment names (RANK / WORLD_SIZE from torch.distributed.run, or the
    single-GPU case).  ("spawn", n): started WITHOUT a launcher for N > 1 -- re-execute under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N`, one rank per GPU, so that a bare
    `python bench.py --gpus 8` measures eight GPUs and never prints a one-GPU number as n_gpus 8.
    ("error", message): the line could not be what it says."""
    if gpus < 1:
        return "error", f"--gpus {gpus}"
    share = env.get("CARL_BENCH_SHARE_GPU") == "1"  # plumbing check of the N > 1 path on a one-GPU box (every rank on GPU 0)
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            return "error", f"--gpus {gpus} but WORLD_SIZE={world}: the line's n_gpus would not be the ranks that ran"
        if world > 1 and not share and n_devices < world:
            return "error", f"--gpus {gpus} but {n_devices} device(s) visible (CARL_BENCH_SHARE_GPU=1 puts every rank on GPU 0)"
        return "run", None
    if gpus == 1:
        return "run", None
    if not share and n_devices < gpus:
        return "error", f"--gpus {gpus} but {n_devices} device(s) visible (CARL_BENCH_SHARE_GPU=1 puts every rank on GPU 0)"
    return "spawn", gpus


def spawn_ranks(n):
    """Re-execute this command line under torch.distributed.run (one process per GPU; rendezvous on 127.0.0.1)."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    what, arg = launch_plan(args.gpus, os.environ, torch.cuda.device_count())
    if what == "error":
        raise SystemExit(arg)
    if what == "spawn":
        raise SystemExit(spawn_ranks(arg))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Plumbing check of the N > 1 path on a ONE-GPU box (tools/bench_2proc_sim.sh): every rank on GPU 0, gloo
    # for the collectives (RCCL refuses two ranks on one device).  Never what the driver runs.
    share_gpu = os.environ.get("CARL_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("CARL_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_dev = device if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1 or args.rccl:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        with _stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            t = torch.zeros(1, device=coll_dev)
            dist.all_reduce(t)  # the communicator is created lazily: here, not inside the timed region
            if coll_dev.type == "cuda":
                torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_over_ranks(x):
        if dist is None:
            return [x]
        t = torch.tensor([x], device=coll_dev, dtype=torch.float64)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        return [float(v.item()) for v in parts]

    _t_last = [time.perf_counter()]

    def tick(label):  # CARL_BENCH_TIMING=1: where the wall clock of a run goes (stderr)
        if os.environ.get("CARL_BENCH_TIMING") == "1" and rank == 0:
            now = time.perf_counter()
            print(f"[bench timing] {label}: {now - _t_last[0]:.1f} s", file=sys.stderr, flush=True)
            _t_last[0] = now

    K, W = args.steps, args.warmup
    T = args.chunk or DEFAULT_CHUNK[args.families[0]]
    n_fam = args.lanes // world if args.strong else args.lanes
    wl = Workload(args.families, n_fam, T, args.buffer_sets, rank, world, device, narrow_actions=args.narrow_actions)
    n = wl.n
    shape = wl.launch_shape()

    # ---- timed region: exactly K fused launches (K x T env steps of every lane), repeated --reps times ----
    regions, med = timed_regions(wl, K, W, args.reps, barrier, max_over_ranks)
    elapsed, avg_launch_s = regions[med]
    roofline = launch_stats(wl, roofline_of(wl, avg_launch_s), K, barrier)
    per_rank_launch_ms = gather_over_ranks(avg_launch_s * 1e3)
    repetitions = {"n": len(regions), "median_index": med, "steps_each": K,
                   "ms_per_step": [w / K * 1e3 for w, _ in regions], "launch_period_ms": [p_ * 1e3 for _, p_ in regions],
                   "value": [n * world * T * K / w for w, _ in regions],
                   "note": "each region = exactly K launches between barrier + synchronize; value / ms_per_step / roofline "
                           "are the median region's"}

    # ---- the same launch train, sustained: the K-launch region above is ~1-2 ms (burst clocks) ----
    import math

    Ks = max(K, int(math.ceil(args.sustained_seconds / max(avg_launch_s, 1e-6))))
    with ClockSampler(device) as clk:
        wall_s, avg_s = wl.train(Ks, 0, barrier)
    el_s = max_over_ranks(wall_s)
    sustained = {"steps": Ks, "seconds": el_s, "value": n * world * T * Ks / el_s, "unit": "env-steps/s",
                 "ms_per_step": el_s / Ks * 1e3, "avg_launch_ms": avg_s * 1e3, "frac": roofline_of(wl, avg_s)["frac"]}
    clocks = clk.record()

    # ---- reporting collective: episodic returns all-gathered over RCCL ------------
    gather_ms, rccl_ranks = None, None
    if dist is not None:
        from carl_amd.distributed import all_gather_episode_stats

        def stats_src():  # RCCL gathers the device vectors in place; the gloo plumbing check stages them
            if backend == "nccl":
                return wl.eng
            return {k: getattr(wl.eng, k).cpu() for k in ("last_return", "last_length", "episodes_done")}

        all_gather_episode_stats(stats_src())  # first call: communicator set-up, untimed
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        stats = all_gather_episode_stats(stats_src())  # an RCCL failure here is fatal: it is the path under test
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        rccl_ranks = int(stats["last_return"].numel() // max(n, 1))
        mean_return = float(stats["last_return"].mean())
    else:
        mean_return = wl.mean_last_return()

    # ---- N > 1: uneven lane shards (a total the ranks do not divide) through the same path: lane_shard gives the first
    # total % world ranks one lane more; every shard takes the staged kernel whatever its lane count (row pitch, ABI 9)
    # and the return all-gather runs in its uneven-counts form (VERDICT r05 "Next" #5 / #6)
    uneven = None
    if dist is not None and world > 1:
        try:
            uneven = uneven_shard_record(args.families[0], args.lanes + 5, rank, world, device, backend, barrier, gather_over_ranks)
        except Exception as e:  # a plumbing side record: it must not take the scaling measurement with it
            uneven = {"error": repr(e)[:300]}

    # ---- per-call path (one launch per env step) -----------------------------------
    per_call = None
    if not args.no_per_call and not wl.mixed:
        import gc

        Kc = 1000
        eng = wl.eng
        a1 = wl.acts[0][0][0].contiguous()
        gc.collect()
        gc.disable()  # (see Workload.train)
        per_call = per_call_record(eng, a1, n * world, Kc, device, world, barrier)
        if world == 1 and dist is None:
            try:
                graph_per_call(per_call, eng, a1, args.families[0], n, Kc, device)
            except Exception as e:  # graph capture is an optimisation of the measurement, not the product
                per_call["graph_error"] = repr(e)[:200]
            if hasattr(eng, "capture_step"):  # the engine's own replayable step (what CARLEnv.step uses for a
                g = eng.capture_step(a1)      # fixed-address action buffer)
                for _ in range(50):
                    g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(Kc):
                    g.replay()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                per_call["captured_value"] = n * Kc / dt
                per_call["captured_ms_per_step"] = dt / Kc * 1e3
                # ... and as a 100-step graph (the form to hold on to when the policy that refills the action buffer is
                # captured too): a ONE-step graph pays a whole graph launch (~11 us) per env step -- slower than the eager
                # `step` (4 us) -- so the eager call is the per-step path and `capture_step(n_steps >= 16)` the replay path
                g100 = eng.capture_step(a1, n_steps=100)
                g100.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(Kc // 100):
                    g100.replay()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                per_call["captured_100_value"] = n * (Kc // 100) * 100 / dt
                per_call["captured_100_ms_per_step"] = dt / ((Kc // 100) * 100) * 1e3
        gc.enable()

    tick("headline + sustained + per_call")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args, args.families[0], wl.tables[0], n_fam)
        except Exception as e:  # the CPU line is a reported baseline: its failure must not take the GPU measurement with it
            _close_pool()
            cpu = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "error": repr(e)[:200]}

    if per_call is not None and rank == 0 and world == 1 and tuple(args.families) == ("cartpole",):
        try:
            per_call["dropin"] = dropin_per_call(device, n_fam, (cpu or {}).get("single_core_value"))
        except Exception as e:  # a reported side record: its failure must not take the measurement with it
            per_call["dropin"] = {"error": repr(e)[:300]}

    tick("cpu_baseline + dropin")
    # ---- the other BASELINE workloads, same launch train ---------------------------
    also = {}
    names = [] if args.also in ("", "none") else args.also.split(",")
    del wl
    torch.cuda.empty_cache()
    for name in names:
        fams, total, mode, chunk = ALSO[name]
        narrow = name.endswith("_u8") or name.endswith("_f16")  # the narrow action formats of the lean staged rollout
        if fams == args.families and chunk in (None, T) and not narrow:
            continue
        split = mode == "strong" or (mode == "follow" and args.strong)
        lanes = total // world if split else total
        mode = "strong" if split else "weak"
        Ta = chunk or DEFAULT_CHUNK[fams[0]]
        fp32 = name.endswith("_fp32")
        w2 = Workload(fams, lanes, Ta, args.buffer_sets, rank, world, device, narrow_actions=narrow, brax_fp32=fp32)
        regs2, m2 = timed_regions(w2, K, W, args.reps, barrier, max_over_ranks)
        el2, avg2 = regs2[m2]
        r2 = launch_stats(w2, roofline_of(w2, avg2), K, barrier)
        also[name] = {
            "workload": " + ".join(f"CARL{f} x {lanes}" for f in fams) + f" contexts/GPU, {Ta} env steps per launch"
                        + (", uint8 actions" if name.endswith("_u8") else ", float16 actions" if narrow else ""),
            "value": w2.n * world * Ta * K / el2, "unit": "env-steps/s", "scaling": mode,
            "lanes_per_gpu": w2.n, "chunk": Ta, "ms_per_step": el2 / K * 1e3,
            "avg_launch_ms": avg2 * 1e3, "frac": r2["frac"], "achieved_GBs": r2["achieved"],
            "bytes_per_unit": r2["bytes_per_unit"], "traffic": r2["traffic"],
            "frac_launch_median": r2["frac_launch_median"], "launch_us": r2["launch_us"],
            "profile_reference": r2.get("profile_reference"),
            "repetitions_ms_per_step": [w / K * 1e3 for w, _ in regs2],
            "mean_last_episode_return": w2.mean_last_return(), "lanes_per_env": w2.launch_shape(),
            "classes": [type(e).__name__ for e in w2.envs],
            "one_launch_pair": (w2.eng.pair_launches > 0) if w2.mixed else None,
        }
        if fp32:
            also[name]["precision"] = ("OPT-IN float32 substeps (CARL_FLAG_BRAX_FP32): brax's own arithmetic under JAX's default; NOT "
                                       "within north_star's 1e-5 of the float64 restatement -- see `deviation`")
            try:
                also[name]["deviation"] = {f: brax_fp32_deviation(f, device) for f in fams}
            except Exception as e:  # a reported side record
                also[name]["deviation"] = {"error": repr(e)[:200]}
        if all(f in BRAX_ENVS for f in fams):
            # the Brax path moves ~1 % of what HBM could: its bound is the vector ALU (committed counters, not measured here)
            also[name]["bound"] = ("vector-ALU issue (roofline_valu: SQ_INSTS_VALU of the full-size launch x cycles per wavefront-"
                                   "instruction / (1 024 SIMDs x 2.4 GHz) against the launch time); `frac` above is the HBM fraction "
                                   "of the same launch, which the metric asks for")
            also[name]["roofline_valu"] = roofline_valu_of(f"{'+'.join(fams)}{'_fp32' if fp32 else ''}:{lanes}:{Ta}", avg2)
            # double-buffered use: the same contexts as two free-running half-batches (one family: two engines of half
            # the lanes; two families: one engine each), launches overlapping across streams
            w4 = SplitWorkload(fams[0], lanes // 2, Ta, args.buffer_sets, rank, world, device) if len(fams) == 1 else w2
            wall4 = max_over_ranks(w4.train_free_running(K, W, barrier))
            also[name]["free_running_two_streams"] = {
                "value": w4.n * world * Ta * K / wall4, "unit": "env-steps/s", "ms_per_step": wall4 / K * 1e3,
                "note": "the same contexts as two half-batches on two HIP streams, joined only at the end of the train "
                        "(double-buffered collection): consecutive launches overlap; not the one-stream figure above"}
            if w4 is not w2:
                del w4
        elif len(fams) > 1:
            # a mixed classic-control batch (BASELINE config 3): each family's launch train on its own stream, joined at
            # the end of the train -- the tail of one family's launch overlaps the head of the other's
            wall4 = max_over_ranks(w2.train_free_running(K, W, barrier))
            also[name]["free_running_two_streams"] = {
                "value": w2.n * world * Ta * K / wall4, "unit": "env-steps/s", "ms_per_step": wall4 / K * 1e3,
                "note": "each family's launch train on its own HIP stream, joined only at the end of the train: consecutive "
                        "launches of the two families overlap; not the one-stream figure above"}
        if name in CPU_BESIDE and rank == 0 and world == 1 and not args.no_cpu_baseline:
            # north_star: every number "next to the reference Python step() timed on the host cores (core count stated)
            # in the same run" -- the restatement of that loop (kind "port"; Brax: the fp64 C restatement of the spring
            # pipeline), same context sets, one record per family and their lane-weighted combination (VERDICT r05 #4;
            # the loop being timed: carl/envs/carl_env.py:321-342)
            also[name]["cpu_baseline"] = cpu_baseline_beside(args, fams, w2.tables, lanes)
        del w2
        torch.cuda.empty_cache()
        tick(f"also.{name}")

    # ---- the 8-GPU shard regime on this ONE GPU ----
    shard8 = None
    if world == 1 and not args.no_shard8:
        shard8 = {}
        full = {k: v["value"] for k, v in also.items()}
        full["+".join(args.families)] = n * T * K / elapsed
        for name, (fams, lanes, full_key) in SHARD8.items():
            Ta = DEFAULT_CHUNK[fams[0]]
            w5 = Workload(fams, lanes, Ta, args.buffer_sets, rank, world, device)
            regs5, m5 = timed_regions(w5, K, W, args.reps, barrier, max_over_ranks)
            el5, avg5 = regs5[m5]
            v5 = w5.n * Ta * K / el5
            v_full = full.get(full_key)
            shard8[name] = {
                "workload": " + ".join(f"CARL{f} x {lanes}" for f in fams) + f" contexts on this GPU (1/8 of the node's), {Ta} env steps per launch",
                "value_per_gpu": v5, "unit": "env-steps/s", "ms_per_step": el5 / K * 1e3, "avg_launch_ms": avg5 * 1e3,
                "lanes_per_gpu": w5.n, "frac": roofline_of(w5, avg5)["frac"], "lanes_per_env": w5.launch_shape(),
                "predicted_node_value": 8 * v5,
                "full_size_1gpu_value": v_full,
                "predicted_speedup_8gpu": (8 * v5 / v_full) if v_full else None,
                "implied_strong_scaling_efficiency": (v5 / v_full) if v_full else None,
                "one_launch_pair": (w5.eng.pair_launches > 0) if w5.mixed else None,
                "repetitions_ms_per_step": [w / K * 1e3 for w, _ in regs5],
            }
            del w5
            torch.cuda.empty_cache()
            tick(f"shard8.{name}")
        shard8["note"] = ("prediction = 8 x this GPU's rate (lanes are independent, no data-path collective: SURVEY 8e); a shard "
                          "this small is bound by ONE wavefront's dependent-issue latency per env step, not by HBM: DESIGN.md 6")

    # ---- N > 1: the OTHER scaling mode of the headline workload, same launch train ----
    other = None
    if world > 1:
        lanes3 = args.lanes if args.strong else args.lanes // world  # headline strong -> weak run; headline weak -> strong run
        w3 = Workload(args.families, lanes3, T, args.buffer_sets, rank, world, device, narrow_actions=args.narrow_actions)
        regs3, m3 = timed_regions(w3, K, W, args.reps, barrier, max_over_ranks)
        el3, avg3 = regs3[m3]
        other = {"scaling": "weak" if args.strong else "strong", "lanes_per_gpu": w3.n, "total_lanes": w3.n * world,
                 "value": w3.n * world * T * K / el3, "unit": "env-steps/s", "ms_per_step": el3 / K * 1e3,
                 "avg_launch_ms": avg3 * 1e3, "frac_per_gpu": roofline_of(w3, avg3)["frac"]}
        del w3
        torch.cuda.empty_cache()

    if rank == 0:
        fam_txt = " + ".join(f"CARL{f} x {n_fam}" for f in args.families)
        line = {
            "metric": "env-steps/sec (whole node) at 65k parallel contexts, 1/2/4/8 MI355X",  # BASELINE.json, verbatim
            "value": n * world * T * K / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            # arithmetic type of the path (state in HBM is float32 throughout): Acrobot's RK4 runs in float64
            "dtype": ("f32" if "acrobot" not in args.families else "f64" if set(args.families) == {"acrobot"} else "f32/f64"),
            "data": "synthetic",
            "config": {"workload": f"{fam_txt} contexts/GPU ({n * world} over the node), StaticSelector lane<->context, auto-reset; one step = one "
                                   f"fused carl_rollout launch of {T} env steps of every lane, full transition written "
                                   f"per env step; {args.buffer_sets} rotating action/output buffer sets",
                       "lanes_per_gpu": n, "total_lanes": n * world, "chunk": T,
                       "env_steps_per_step": n * world * T, "buffer_sets": args.buffer_sets,
                       "parallelism": f"lane-shard x{world}", "lanes_per_env": shape},
            "bench_version": BENCH_VERSION, "workload_id": f"{'+'.join(args.families)}{'_narrow' if args.narrow_actions else ''}:{n}:{T}:v{BENCH_VERSION}",
            # the figure comparable with BENCH_r01 - r03 (same workload, 250-step launches: the headline definition of those
            # rounds; r04 moved `value` to SURVEY 8d's 1 000-step launches) -- ADVICE r04
            "value_r03_definition": (also.get("cartpole_T250") or {}).get("value"),
            "repetitions": repetitions, "clocks": clocks, "shard8": shard8,
            "roofline": roofline, "cpu_baseline": cpu, "sustained": sustained,
            ("weak" if args.strong else "strong"): other, "per_call": per_call, "also": also,
            "mean_last_episode_return": mean_return, "return_allgather_ms": gather_ms, "rccl_ranks": rccl_ranks,
            "collective_backend": (backend if dist is not None else None),
            "per_rank_avg_launch_ms": per_rank_launch_ms, "uneven_shards": uneven,
        }
        print(json.dumps(line), flush=True)
    _close_pool()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
