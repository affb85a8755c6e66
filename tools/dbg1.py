import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import lipmpc, lipmpc_oracle as O
from helpers import closed_loop_problems
from test_gpu_parity import run_gpu
probs=list(closed_loop_problems(8,10,6,25,seed=108))
np.set_printoptions(precision=4, linewidth=200)
for sel in ([64],[65],[66],[67],[64,65,66,67],[63,64],[64,64,64,64],[64,0,1,2]):
    res=run_gpu([probs[i] for i in sel],8,10,5)
    print(sel,'status',res['status'],'iters',res['iters'],'diag',res['diag'].tolist())
