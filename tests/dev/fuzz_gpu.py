"""Dev tool: fuzz the step kernel against the C oracle with arbitrary (mostly infeasible or odd) inputs."""
import sys, os, numpy as np, torch, time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import lipmpc, c_oracle
from importlib import import_module
synth=import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
for (N,n_obs,B) in [(8,10,8192),(3,3,8192),(1,0,2048),(1,5,2048),(2,9,4096),(4,14,4096),(4,0,1024),(5,6,4096),(8,13,4096),(12,10,2048),(12,13,1024),(16,30,512),(16,50,1024)]:
    xy,nv=synth.synthetic_fields(64,max(n_obs,1),0.5,9.5,(0,0),(10,10),seed=int(rng.integers(1e6)))
    xy,nv=xy[:,:n_obs],nv[:,:n_obs]
    idx=rng.integers(0,64,B); xy=xy[idx].copy(); nv=nv[idx].copy()
    nv[rng.random(nv.shape)<0.1]=0                                  # empty slots
    deg=rng.random(B)<0.02
    if n_obs: xy[deg,0,1]=xy[deg,0,0]                              # zero-length edges
    st=np.zeros((B,5)); st[:,0]=rng.uniform(0,10,B); st[:,2]=rng.uniform(0,10,B)
    st[:,1]=rng.normal(0,0.3,B); st[:,3]=rng.normal(0,0.3,B); st[:,4]=rng.uniform(-4,4,B)
    calm=rng.random(B)<0.5; st[calm,1]*=0.1; st[calm,3]=np.where(rng.random(calm.sum())<0.5,0.25,-0.25)
    goal=rng.uniform(-2,12,(B,2)); foot=rng.choice([-1,1],B).astype(np.int8); delta=np.where(rng.random(B)<0.5,0.0,rng.uniform(0,0.5,B))
    for flags in (0, lipmpc.FLAG_NO_PRESOLVE):      # presolve + smallest solver body / every row in the handle's own body
        P=lipmpc.LipMpcParams(N=N,n_obs_max=n_obs,v_max=5,flags=flags); sv=lipmpc.BatchedLipMpc(P)
        dev=lambda a,dt: torch.as_tensor(np.ascontiguousarray(a),dtype=dt,device="cuda")
        t=time.time()
        out=sv.plan_step_batch(dev(st,torch.float64),dev(goal,torch.float64),dev(foot,torch.int8),dev(xy,torch.float64) if n_obs else None,dev(nv,torch.int32) if n_obs else None,dev(delta,torch.float64))
        torch.cuda.synchronize(); tg=time.time()-t
        ref=c_oracle.plan_step_batch(P,st,goal,foot,xy if n_obs else None,nv if n_obs else None,delta,n_threads=16)
        gs=out["status"].cpu().numpy(); same=gs==ref["status"]
        ok=same&(gs==0); U=out["U"].cpu().numpy()
        du=np.abs(U[ok]-ref["U"][ok]).max() if ok.any() else 0
        nanbad=np.isnan(U[ok]).any()
        act_diff=int(np.any(out["active"].cpu().numpy().view(np.uint64)[ok]!=ref["active"][ok],axis=1).sum())     # tight sets, every certified problem
        print(f"N={N} n_obs={n_obs} B={B} flags={flags}{' split' if getattr(sv,'_ws',None) is not None else ''}: gpu {tg*1e3:.1f} ms | status gpu {np.bincount(gs,minlength=5).tolist()} oracle {np.bincount(ref['status'],minlength=5).tolist()} | mismatches {int((~same).sum())} | max dU {du:.2e} | active (tight set) differs on {act_diff} of {int(ok.sum())} | max iters {int(out['iters'].max())} nan {nanbad}")
