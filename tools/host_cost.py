"""Host-side cost of the two launch paths (run on the GPU box): fused rollout enqueue, per-call step
(eager / fast path / captured graph replay), and the pieces of an eager step."""
import sys
import time

import torch

sys.path.insert(0, ".")
from carl_amd.context.selection import StaticSelector  # noqa: E402
from carl_amd.envs import CARLPendulum  # noqa: E402

n = 65536
env = CARLPendulum(num_envs=n, device="cuda:0", context_selector=StaticSelector)
eng = env.env
env.reset(seed=0)
T = 250
a = torch.rand((T, n), device="cuda:0") * 4 - 2
out = eng.alloc_rollout(T)
for _ in range(20):
    eng.rollout(a, out)
torch.cuda.synchronize()
for reps in (50, 200):
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.rollout(a, out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rollout x{reps}: host enqueue {1e6 * (t1 - t0) / reps:.1f} us/launch, total {1e6 * (t2 - t0) / reps:.1f} us/launch")


def timeit(name, fn, reps=2000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:44s} host {1e6 * (t1 - t0) / reps:6.2f} us/call   incl. drain {1e6 * (t2 - t0) / reps:6.2f} us/call")


a1 = a[0].contiguous()
timeit("eng.step(same tensor)  [fast path]", lambda: eng.step(a1))
timeit("eng.step(new view each call) [validated]", lambda: eng.step(a[0]))
timeit("CARLEnv.step(same tensor)", lambda: env.step(a1))
g1 = eng.capture_step(a1)
timeit("capture_step(n=1).replay()", g1.replay)
g8 = eng.capture_step(a1, 8)
timeit("capture_step(n=8).replay()  (per 8 steps)", g8.replay, 500)
from carl_amd.engine import _current_device, _raw_stream  # noqa: E402

timeit("  _raw_stream", lambda: _raw_stream(0))
timeit("  _current_device", _current_device)
timeit("  torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
timeit("  ctypes carl_step only", lambda: eng._c_step_fast(_raw_stream(0)))
