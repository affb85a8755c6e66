import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import lipmpc
d=np.load(os.path.join(ROOT,'tests/golden/lidar_golden.npz'))
rings=[d["env"][0][j][:d["env_nv"][0][j]] for j in range(20)]
B=4096; rng=np.random.default_rng(0); pos=rng.uniform(-0.8,5.8,(B,2)); st=np.zeros((B,5)); st[:,0]=pos[:,0]; st[:,2]=pos[:,1]
d_st=torch.as_tensor(st,device="cuda"); sensor=lipmpc.LidarSensor(rings,lidar_range=1.5,n_obs_max=12,v_max=32)
noise=0.01*torch.randn((B,360,2),dtype=torch.float64,device="cuda")
for _ in range(3): sensor.sense(d_st,noise)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): sensor.sense(d_st,noise)
e1.record(); torch.cuda.synchronize(); print('stop',os.environ.get('LIPMPC_LIDAR_STOP','0'),'%.3f ms'%(e0.elapsed_time(e1)/10))
