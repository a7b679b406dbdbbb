"""Per-call `carl_step` under hipGraph replay: launch-to-launch PERIOD (HIP events around the replays) against the
per-dispatch DURATION rocprofv3 reports for the same kernel (2.41 us period vs 4.03 us
"duration" -- a kernel cannot be longer than its launch period unless consecutive dispatches overlap).
    python tools/per_call_period.py                      # prints the period
    rocprofv3 --kernel-trace --output-format csv -d DIR -o pc -- python tools/per_call_period.py
    python tools/per_call_period.py --trace DIR          # start-to-start deltas and durations from the trace"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(family="pendulum", n=65536, steps=100, reps=50):
    import torch

    sys.argv = [sys.argv[0]]
    import bench

    dev = torch.device("cuda", 0)
    env, _ = bench.make_env(family, n, 0, 1, dev)
    env.reset(seed=0)
    eng = env.env
    a = bench.make_actions(eng, 1, dev, 1)[0].contiguous()
    for _ in range(20):
        eng.step(a)
    g = eng.capture_step(a, steps)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{family} x {n}: graph of {steps} step launches, {reps} replays: period {e0.elapsed_time(e1) * 1e3 / (reps * steps):.3f} us per step", flush=True)
    e0.record()
    for _ in range(2000):
        eng.step(a)
    e1.record()
    torch.cuda.synchronize()
    print(f"eager: {e0.elapsed_time(e1) * 1e3 / 2000:.3f} us per step", flush=True)


def trace(root):
    rows = []
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    import statistics as S

    dur = [e - s for s, e in zip(st, en)]
    gap = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    tight = [(g, d) for g, d in zip(gap, dur) if g < 20000]  # consecutive dispatches of one train
    print(f"{len(rows)} step_kernel dispatches; duration median {S.median(dur) / 1e3:.3f} us, mean {S.mean(dur) / 1e3:.3f} us")
    print(f"start-to-start of consecutive dispatches: median {S.median(g for g, _ in tight) / 1e3:.3f} us, "
          f"share of dispatches that START before their predecessor's END timestamp: "
          f"{sum(1 for i in range(len(st) - 1) if st[i + 1] < en[i]) / max(len(st) - 1, 1):.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--trace":
        trace(sys.argv[2])
    else:
        run()
