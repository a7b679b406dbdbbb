import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from carl_amd.envs.brax.models import ant_sys
from carl_amd.brax_engine import BraxVecEngine
from oracle import brax as B, oracle as O
NAMES=["gravity","friction","elasticity","ang_damping","mass_torso","viscosity","target_distance","target_direction","target_radius"]
DEFAULT=np.array([-9.8,1.0,0.0,-0.05,10.0,0.0,100.0,1.0,5.0])
def rel(g,w): g=np.asarray(g,np.float64); w=np.asarray(w,np.float64); return np.abs(g-w)/(1+np.abs(w))
for nf in (1,10):
    s=ant_sys(NAMES); s.n_frames=nf
    rng=np.random.default_rng(1); n=4096
    rows=np.tile(DEFAULT,(n,1)); rows[:,0]=rng.uniform(-15,-5,n); rows[:,1]=rng.uniform(.3,1.5,n); rows[:,4]=rng.uniform(5,15,n)
    rows=rows.astype(np.float32).astype(np.float64)
    kw=dict(selector=O.SEL_STATIC,seed=5,ctx_idx0=np.arange(n))
    eng=BraxVecEngine(s,len(NAMES),rows,n,"cuda",max_episode_steps=1000,auto_reset=False,**kw)
    ora=B.Engine(s,rows,n,max_steps=1000,autoreset=False,**kw)
    eng.reset(); ora.reset()
    errs=[]; errr=[]
    for t in range(60 if nf==10 else 300):
        ora.state[:]=eng.state_np()
        a=rng.uniform(-1.2,1.2,(n,8)).astype(np.float32)
        obs,rew,term,trunc=eng.step(torch.as_tensor(a)); out=ora.step(a)
        errs.append(rel(obs.cpu().numpy(),out.obs).max(1)); errr.append(rel(rew.cpu().numpy(),out.reward))
        es=rel(eng.state_np(), ora.state)
    e=np.concatenate(errs); r=np.concatenate(errr)
    print("n_frames",nf,"obs err pct 50/99/99.9/max: %.2e %.2e %.2e %.2e"%tuple(np.percentile(e,[50,99,99.9,100])), " reward: %.2e %.2e %.2e %.2e"%tuple(np.percentile(r,[50,99,99.9,100])), "state max %.2e"%es.max())
    # which obs column is worst
    col=np.stack([rel(obs.cpu().numpy(),out.obs)]).max(0).max(0); print(" worst cols", np.argsort(col)[-5:], np.sort(col)[-5:])
