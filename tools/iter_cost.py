"""Dev tool: time per interior-point iteration against the number of LDCBF row slots per lane (uniform batch,
interior flag, max_iter 3 vs 8) -> marginal cost of a row slot and of the slot-independent part."""
import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import lipmpc
from importlib import import_module
synth=import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev=torch.device("cuda",0); B=4096
def run(N,n_obs):
    xy,nv=synth.synthetic_fields(8,max(n_obs,1),0.5,9.5,(0,0),(10,10),seed=1)
    obs_xy=torch.as_tensor(np.repeat(xy[:1,:n_obs],B,0),device=dev).contiguous() if n_obs else None
    obs_nv=torch.as_tensor(np.repeat(nv[:1,:n_obs],B,0),device=dev).contiguous() if n_obs else None
    goal=torch.tensor([[10.,10.]],dtype=torch.float64,device=dev).repeat(B,1).contiguous()
    state=torch.zeros((B,5),dtype=torch.float64,device=dev); foot=torch.ones((B,),dtype=torch.int8,device=dev)
    ts={}
    for mi in (3,8):
        sv=lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N,n_obs_max=n_obs,v_max=5,flags=1,max_iter=mi),0); out=sv.alloc_outputs(B)
        for _ in range(3): sv.plan_step_batch(state,goal,foot,obs_xy,obs_nv,None,out=out)
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): sv.plan_step_batch(state,goal,foot,obs_xy,obs_nv,None,out=out)
        e1.record(); torch.cuda.synchronize(); ts[mi]=e0.elapsed_time(e1)/30*1e3
        assert int(out['iters'][0])==mi, int(out['iters'][0])
    print(f"N={N} n_obs={n_obs}: per-iteration {(ts[8]-ts[3])/5:.3f} us, fixed {ts[3]-3*(ts[8]-ts[3])/5:.1f} us")
for N,n_obs in ((8,0),(8,4),(8,10),(8,12),(8,14),(8,16),(8,26),(16,0),(16,10),(16,14),(16,50)): run(N,n_obs)
