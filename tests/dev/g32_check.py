"""Dev tool: G=32 (N=12, 10 obstacles) statuses / iterations / finish rounds against the C oracle."""
import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle')); sys.path.insert(0,os.path.join(ROOT,'tests'))
import lipmpc, c_oracle
from helpers import closed_loop_problems
N,n_obs=12,10
probs=list(closed_loop_problems(N,n_obs,8,25,seed=5))
P=lipmpc.LipMpcParams(N=N,n_obs_max=n_obs,v_max=5); sv=lipmpc.BatchedLipMpc(P)
st=np.array([p[0] for p in probs]); goal=np.array([p[1] for p in probs],float); foot=np.array([p[2] for p in probs],np.int8)
xy,nv=lipmpc.pack_rings([p[3] for p in probs],n_obs,5)
dev=lambda a,dt: torch.as_tensor(np.ascontiguousarray(a),dtype=dt,device="cuda")
out=sv.plan_step_batch(dev(st,torch.float64),dev(goal,torch.float64),dev(foot,torch.int8),dev(xy,torch.float64),dev(nv,torch.int32),None,with_diag=True)
torch.cuda.synchronize(); g={k:v.cpu().numpy() for k,v in out.items()}
ref=c_oracle.plan_step_batch(P,st,goal,foot,xy,nv,None,n_threads=8)
mm=np.where(g["status"]!=ref["status"])[0]
print(len(probs),'status gpu',np.bincount(g["status"],minlength=5),'ref',np.bincount(ref["status"],minlength=5),'mismatch',len(mm))
for b in mm[:8]: print(b,'gpu',g["status"][b],g["iters"][b],np.round(g["diag"][b],12).tolist(),'| ref',ref["status"][b],ref["iters"][b],np.round(ref["diag"][b],12).tolist())
okb=(g["status"]==0)&(ref["status"]==0); print('max dU',np.abs(g["U"][okb]-ref["U"][okb]).max(), 'rounds gpu hist',np.bincount(g["diag"][:,0].astype(int)),'ref',np.bincount(ref["diag"][:,0].astype(int)))
print('iters differ at',np.where(g["iters"]!=ref["iters"])[0][:10], 'rounds differ at', np.where(g["diag"][:,0]!=ref["diag"][:,0])[0][:10])
d=np.where(g["iters"]!=ref["iters"])[0]
print('gpu iters',g["iters"][d],'ref iters',ref["iters"][d],'sum gpu',g["iters"].sum(),'sum ref',ref["iters"].sum())
print('max dX all ok',np.abs(g["X"][okb]-ref["X"][okb]).max(),'obj rel',np.abs(g["obj"][okb]-ref["obj"][okb]).max())
