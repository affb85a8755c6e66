import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import lipmpc, lipmpc_oracle as O
from helpers import closed_loop_problems
from test_gpu_parity import run_gpu
N=16
probs=list(closed_loop_problems(N,50,1,3,seed=N))
for keep in [10,26,30,40,50]:
    pp=[(p[0],p[1],p[2],p[3][:keep],p[4]) for p in probs]
    res=run_gpu(pp,N,50,5)
    r=[O.plan_step(*p,O.Params(N=N)) for p in pp]
    print('present',keep,'gpu status',res['status'],'iters',res['iters'],'oracle',[x['status'] for x in r],[x['iters'] for x in r],'diag0',res['diag'][0])
