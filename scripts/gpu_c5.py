"""A few sweeps of the C5-shaped constrained chain (P = 256) for `rocprofv3 --kernel-trace --stats`: the per-kernel
split between the X pass and the replicated beta stage.   BL_N=4000000 python scripts/gpu_c5.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bayeslogit_amd import device as D
sys.argv=['x']
import bench
dev=torch.device('cuda:0')
N,P=int(os.environ.get('BL_N','4000000')),int(os.environ.get('BL_P','256'))
X,y,bt=bench.synth_logit(D,dev,N,P)
nn=torch.ones(N,dtype=torch.float64,device=dev)
sh=D.GibbsShard(X,y,nn,seed=20240004)
sh.set_prior(np.zeros(P),np.eye(P)*0.01); sh.set_bp_local(); sh.finish_bp(); sh.set_beta(np.zeros(P))
import hashlib, time
dig=hashlib.sha1(); ms=[]
for s in range(int(os.environ.get('BL_SWEEPS','12'))):
    sh.sweep_local(s,None)
    torch.cuda.synchronize(); t0=time.perf_counter()
    sh.draw_beta(s,1)
    torch.cuda.synchronize(); ms.append((time.perf_counter()-t0)*1e3)
    dig.update(sh.beta().cpu().numpy().tobytes())
    print("sweep",s, "min beta", float(sh.beta()[:-1].min()), "beta stage ms %.3f"%ms[-1], flush=True)
print("beta stage median ms %.3f  digest %s"%(float(np.median(ms[2:])),dig.hexdigest()[:12]))
D.sync_status()
