"""A/B of two builds of the P <= 64 beta stage: hashes of the beta history of short chains at several P, both draws
(the dense routines of round 3 are meant to reproduce round 2's bits), and the event-timed beta stage at P = 64.
    [BAYESLOGIT_LIB=variant.so] python scripts/gpu_beta64_ab.py        (run once per build, compare the lines)"""
import hashlib
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
for P in (1, 2, 5, 8, 13, 24, 37, 60, 63, 64):
    N = 4000 + 100 * P
    X, y, bt = bench.synth_logit(D, dev, N, P)
    nn = torch.ones(N, dtype=torch.float64, device=dev)
    for con in (0, 1):
        sh = D.GibbsShard(X, y, nn, seed=77 + P)
        sh.set_prior(np.zeros(P), np.eye(P) * 0.01)
        sh.set_bp_local()
        sh.finish_bp()
        sh.set_beta(np.full(P, 0.05) if con else np.zeros(P))
        hist = []
        for s in range(6):
            sh.sweep_local(s, None)
            sh.draw_beta(s, con)
            hist.append(sh.get_beta().copy())
        D.sync_status()
        h = np.ascontiguousarray(np.stack(hist))
        print(f"{tag:20s} P {P:3d} constrain {con} beta sha {hashlib.sha256(h.tobytes()).hexdigest()[:16]} last {h[-1, 0]:.17g}")

# chains pressed against their bounds (few rows per coefficient: most moves need their bounds, the scans go move by move and
# come back to speculating every eighth scan) and in between
for N, P in ((2560, 64), (6400, 64), (20000, 64), (1500, 50)):
    X, y, bt = bench.synth_logit(D, dev, N, P)
    nn = torch.ones(N, dtype=torch.float64, device=dev)
    sh = D.GibbsShard(X, y, nn, seed=5 + N)
    sh.set_prior(np.zeros(P), np.eye(P) * 0.01)
    sh.set_bp_local()
    sh.finish_bp()
    sh.set_beta(np.full(P, 0.02))
    hist = []
    for s in range(10):
        sh.sweep_local(s, None)
        sh.draw_beta(s, 1)
        hist.append(sh.get_beta().copy())
    D.sync_status()
    h = np.ascontiguousarray(np.stack(hist))
    print(f"{tag:20s} pressed N {N:6d} P {P:3d} beta sha {hashlib.sha256(h.tobytes()).hexdigest()[:16]} min beta {h[-1, :-1].min():.3e}")

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
    for s in range(24):
        sh.sweep_local(s, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sh.draw_beta(s, con)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    D.sync_status()
    print(f"{tag:20s} P 64 constrain {con} beta stage ms: min {min(ms[4:]):.4f} median {np.median(ms[4:]):.4f}")
