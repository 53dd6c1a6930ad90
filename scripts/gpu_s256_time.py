"""Event-timed P = 256 sweep on the C5 shard (12.5e6 x 256) at beta = 0 and at the generating beta: ms per sweep_local.
    [BAYESLOGIT_LIB=variant.so] python scripts/gpu_s256_time.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bayeslogit_amd import device as D

sys.argv = ['x']
import bench

dev = torch.device('cuda:0')
N, P = int(os.environ.get('BL_N', '12500000')), 256
X, y, bt = bench.synth_logit(D, dev, N, P)
nn = torch.ones(N, dtype=torch.float64, device=dev)
sh = D.GibbsShard(X, y, nn, seed=20240004)
for name, b in (("beta = 0", np.zeros(P)), ("beta = truth", bt.cpu().numpy())):
    sh.set_beta(b)
    ms = []
    for s in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sh.sweep_local(s, None)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    D.sweep_deferred_rows()
    sh.sweep_local(99, None)
    torch.cuda.synchronize()
    print(f"{os.environ.get('BAYESLOGIT_LIB', 'default')[-24:]:24s} {name:14s} sweep ms min {min(ms[2:]):.3f} "
          f"median {np.median(ms[2:]):.3f}  deferred rows {D.sweep_deferred_rows()}")
D.sync_status()
