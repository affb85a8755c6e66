"""Dev tool: status / iteration statistics of the config-5 step (bench.py's scenario: 4096 robots on one crowded map)."""
import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import lipmpc
from importlib import import_module
synth=import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev=torch.device("cuda",0); B,N=4096,3
exy,env=synth.synthetic_fields(1,20,-1.0,6.0,(-5.0,-5.0),(50.0,50.0),seed=9,delta=0.6)
rings=[exy[0,j,:env[0,j]] for j in range(20) if env[0,j]>0]
sensor=lipmpc.LidarSensor(rings,lidar_range=1.5,resolution=360,n_obs_max=12,v_max=32,device=0)
gen=torch.Generator(device=dev).manual_seed(3)
pos=torch.rand((B,2),dtype=torch.float64,device=dev,generator=gen)*7.0-1.0
state=torch.zeros((B,5),dtype=torch.float64,device=dev); state[:,0]=pos[:,0]; state[:,2]=pos[:,1]
noise=0.01*torch.randn((B,360,2),dtype=torch.float64,device=dev,generator=gen)
goal=torch.tensor([[5.0,5.0]],dtype=torch.float64,device=dev).repeat(B,1).contiguous()
foot=torch.ones((B,),dtype=torch.int8,device=dev)
sen=sensor.sense(state,noise)
for flags in (0,1):
    sv=lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N,n_obs_max=12,v_max=32,flags=flags),0)
    o=sv.alloc_outputs(B,with_diag=True)
    for _ in range(3): sv.plan_step_batch(state,goal,foot,sen["obs_xy"],sen["obs_nv"],None,out=o)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): sv.plan_step_batch(state,goal,foot,sen["obs_xy"],sen["obs_nv"],None,out=o)
    e1.record(); torch.cuda.synchronize()
    st=o["status"].cpu().numpy(); it=o["iters"].cpu().numpy(); r=o["diag"][:,0].cpu().numpy().astype(int)
    print('flags',flags,'ms',e0.elapsed_time(e1)/10,'status',np.bincount(st,minlength=5).tolist(),'iters mean %.1f max %d'%(it.mean(),it.max()),'iters hist',np.bincount(it).tolist(),'rounds',np.bincount(r).tolist())
    for s in range(5):
        if (st==s).any(): print('   status',s,'iters max',it[st==s].max(),'mean %.1f'%it[st==s].mean())
b=int(np.where(st==1)[0][0]) if (st==1).any() else -1
if b>=0:
    np.savez(os.path.join(ROOT,'gpurun_out','cfg5_maxiter.npz'), state=state[b].cpu().numpy(), goal=goal[b].cpu().numpy(), obs_xy=sen["obs_xy"][b].cpu().numpy(), obs_nv=sen["obs_nv"][b].cpu().numpy())
    print('saved problem',b)
