"""Dev tool: cost split of one launch on a uniform batch (4096 copies of one problem): front end, IPM, finish."""
import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import lipmpc
from importlib import import_module
synth=import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev=torch.device("cuda",0); B=4096; N=8; n_obs=10
xy,nv=synth.synthetic_fields(8,n_obs,0.5,9.5,(0,0),(10,10),seed=1)
obs_xy=torch.as_tensor(np.repeat(xy[:1],B,0),device=dev); obs_nv=torch.as_tensor(np.repeat(nv[:1],B,0),device=dev)
goal=torch.tensor([[10.,10.]],dtype=torch.float64,device=dev).repeat(B,1).contiguous()
walker=lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N,n_obs_max=n_obs,v_max=5,flags=1),0)
state,foot=synth.walk_states(walker,obs_xy,obs_nv,goal,12,seed=3)
state=state[:1].repeat(B,1).contiguous(); foot=foot[:1].repeat(B).contiguous()
def t(P,label):
    sv=lipmpc.BatchedLipMpc(P,0); out=sv.alloc_outputs(B,with_diag=True)
    for _ in range(3): sv.plan_step_batch(state,goal,foot,obs_xy,obs_nv,None,out=out)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): sv.plan_step_batch(state,goal,foot,obs_xy,obs_nv,None,out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1)/30*1e3:8.1f} us  iters {int(out['iters'][0])} rounds {int(out['diag'][0,0])} status {int(out['status'][0])}")
t(lipmpc.LipMpcParams(N=N,n_obs_max=n_obs,v_max=5),"full (IPM + finish)")
t(lipmpc.LipMpcParams(N=N,n_obs_max=n_obs,v_max=5,flags=1),"interior (no finish)")
for mi in (1,2,4,8):
    t(lipmpc.LipMpcParams(N=N,n_obs_max=n_obs,v_max=5,flags=1,max_iter=mi),f"interior, max_iter={mi}")
