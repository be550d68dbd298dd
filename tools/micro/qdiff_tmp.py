import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from kinpoly_amd import sim as kp
import test_gpu_parity as tp
n=300
qpos,qvel=tp.make_states(n,41,lift=0.0,vel=0.5,noise=0.2)
act=np.random.default_rng(42).normal(size=(n,75))*0.2
for steps in (1,3):
    ref,dref=tp._run_sched(kp,kp.KpModel(substeps_per_job=0),n,qpos,qvel,act,steps=steps)
    for spj,split in ((5,None),(15,None),(0,[5,5,5]),(0,[15])):
        got,dg=tp._run_sched(kp,kp.KpModel(substeps_per_job=spj,queue_slots=48),n,qpos,qvel,act,steps=steps,split=split)
        d=[np.abs(a.astype(np.float64)-b).max() for a,b in zip(ref,got)]
        ne=[int((a!=b).any(1).sum()) for a,b in zip(ref,got)]
        print('steps',steps,'spj',spj,'split',split,'maxdiff',d,'envs differing',ne,'diag eq',bool((dref==dg).all()))
