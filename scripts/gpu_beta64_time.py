"""Event-timed P = 64 beta stage (both draws) on a C4-shaped posterior, for A/B runs of two builds on the same box:
    for l in a.so b.so; do BAYESLOGIT_LIB=$l python scripts/gpu_beta64_time.py; done"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bayeslogit_amd import device as D

sys.argv = ['x']
import bench

dev = torch.device('cuda:0')
tag = os.environ.get('BAYESLOGIT_LIB', 'default')[-20:]
N, P = 2000000, 64
X, y, bt = bench.synth_logit(D, dev, N, P)
nn = torch.ones(N, dtype=torch.float64, device=dev)
for con in (0, 1):
    sh = D.GibbsShard(X, y, nn, seed=20240004)
    sh.set_prior(np.zeros(P), np.eye(P) * 0.01)
    sh.set_bp_local()
    sh.finish_bp()
    sh.set_beta(np.zeros(P))
    ms = []
    for s in range(104):
        sh.sweep_local(s, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sh.draw_beta(s, con)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    D.sync_status()
    m = np.array(ms[4:])
    print(f"{tag:20s} P 64 constrain {con} beta stage ms over {m.size} draws: min {m.min():.4f} median {np.median(m):.4f} mean {m.mean():.4f} max {m.max():.4f}")
