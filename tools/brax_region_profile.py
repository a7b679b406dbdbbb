"""Where a Brax step launch spends its time, by region (s_memtime clocks inside the kernel; measurement build only).

    BRAX_ONLY=1 tools/build_variant.sh prof -DCARL_BRAX_PROFILE
    CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_prof.so CARL_AMD_NO_BUILD=1 python tools/brax_region_profile.py [ant humanoid ...]

Per family x 32 768 envs (``--lanes``), 20-step launches: the share of every region of brax_kernels.hip.h::run (MODE 1)
in the wavefronts' summed clock time, and the mean per wavefront-substep in shader clocks."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from carl_amd import _lib  # noqa: E402
from carl_amd import envs as E  # noqa: E402
from carl_amd.brax_engine import BraxVecEngine  # noqa: E402
from carl_amd.envs.brax.models import SYSTEMS  # noqa: E402

REGIONS = ["load", "prologue", "joints(A)", "bodies(B)", "epilogue", "observe", "reward", "done/reset", "output", "store",
           "B:sum+euler", "B:contacts"]
FAM = {"ant": "CARLBraxAnt", "halfcheetah": "CARLBraxHalfcheetahStiffness", "humanoid": "CARLBraxHumanoidStiffness",
       "hopper": "CARLBraxHopper", "walker2d": "CARLBraxWalker2d", "pusher": "CARLBraxPusher"}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lanes = 32768
    width = 0
    for a in sys.argv[1:]:
        if a.startswith("--lanes="):
            lanes = int(a.split("=")[1])
        if a.startswith("--width="):
            width = int(a.split("=")[1])
    lib = _lib.load()
    read = lib.carl_brax_profile_read
    read.restype = C.c_int
    read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 32)()
    T = 20
    for fam in args or ["ant", "halfcheetah", "humanoid"]:
        cls = getattr(E, FAM[fam])
        feats = cls.get_context_features()
        names = list(feats)
        rng = np.random.default_rng(5)
        rows = np.tile([float(f.default_value) for f in feats.values()], (lanes, 1))
        for k, (lo, hi) in {"gravity": (-15, -5), "friction": (0.3, 1.5), "joint_stiffness": (0.5, 2.0)}.items():
            if k in names:
                rows[:, names.index(k)] = rng.uniform(lo, hi, lanes)
        s = SYSTEMS[cls.env_name](names)
        if width:
            s.lanes_per_env = width
        eng = BraxVecEngine(s, len(names), rows, lanes, "cuda", selector=0, ctx_idx0=np.arange(lanes), seed=3,
                            max_episode_steps=1000)
        eng.reset()
        amp = float(max(s.act_hi[: s.n_act]))
        acts = torch.as_tensor(rng.uniform(-amp, amp, (T, lanes, s.n_act)).astype(np.float32), device="cuda")
        out = eng.alloc_rollout(T)
        for _ in range(3):
            eng.rollout(acts, out=out)
        torch.cuda.synchronize()
        read(buf, 1)
        reps = 5
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            eng.rollout(acts, out=out)
        ev1.record()
        torch.cuda.synchronize()
        assert read(buf, 1) == len(REGIONS)
        tot = float(sum(buf[k] for k in range(len(REGIONS))))
        waves = buf[len(REGIONS)]
        substeps = reps * T * int(s.n_frames) * ((lanes + (64 // eng_width(eng, s)) - 1) // (64 // eng_width(eng, s)))
        print(f"{fam}: {lanes} envs, {T}-step launches {ev0.elapsed_time(ev1) / reps:.3f} ms (instrumented), "
              f"{waves / reps:.0f} wavefronts per launch, {tot / substeps:.0f} clocks per wavefront-substep (all regions)")
        for k, name in enumerate(REGIONS):
            print(f"    {name:12s} {100.0 * buf[k] / tot:5.1f} %   {buf[k] / substeps:7.0f} clocks per wavefront-substep")


def eng_width(eng, s):
    return int(s.lanes_per_env) if int(s.lanes_per_env) > 0 else max(int(s.n_links), 1)


if __name__ == "__main__":
    main()
