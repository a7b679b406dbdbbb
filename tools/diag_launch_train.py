"""Where does a 200-launch train of the CartPole rollout lose time?  Host timestamps per launch + one HIP-event pair per
20 launches.  Run on the GPU box."""
import sys, time
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
import torch
import bench

dev = torch.device("cuda", 0)
for fam in ("cartpole", "pendulum"):
    wl = bench.Workload((fam,), 65536, 250, 2, 0, 1, dev)
    for _ in range(20):
        wl.launch()
    torch.cuda.synchronize()
    K = 200
    host = []
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K // 20 + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(K):
        h0 = time.perf_counter()
        wl.launch()
        host.append(time.perf_counter() - h0)
        if (k + 1) % 20 == 0:
            evs[(k + 1) // 20].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    seg = [evs[i].elapsed_time(evs[i + 1]) / 20 for i in range(K // 20)]
    host.sort()
    print(fam, "wall per launch %.1f us" % (wall / K * 1e6), "| GPU ms per launch by 20-launch segment:", [round(s, 4) for s in seg],
          "| host call us: median %.1f p90 %.1f max %.1f" % (host[K // 2] * 1e6, host[int(K * 0.9)] * 1e6, host[-1] * 1e6), flush=True)
