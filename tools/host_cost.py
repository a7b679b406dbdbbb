import time, torch, sys
sys.path.insert(0,'.')
from carl_amd.envs import CARLPendulum
from carl_amd.context.selection import StaticSelector
n=65536
env=CARLPendulum(num_envs=n, device="cuda:0", context_selector=StaticSelector)
eng=env.env; env.reset(seed=0)
T=250
a=torch.rand((T,n),device="cuda:0")*4-2
out=eng.alloc_rollout(T)
for _ in range(20): eng.rollout(a,out)
torch.cuda.synchronize()
for reps in (50,200,800):
    t0=time.perf_counter()
    for _ in range(reps): eng.rollout(a,out)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(f"{reps} launches: host enqueue {1e6*(t1-t0)/reps:.1f} us/launch, total {1e6*(t2-t0)/reps:.1f} us/launch")
