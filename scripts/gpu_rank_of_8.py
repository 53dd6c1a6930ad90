"""What one rank of an 8-GPU run of C4 does per sweep (N / 8 = 1.25e6 rows, P = 64), timed on one GPU: the local X pass, the
constrained beta stage, and the two back to back as the driver issues them (the P x P all-reduce between them is the one
thing missing).    python scripts/gpu_rank_of_8.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bayeslogit_amd import device as D

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
sys.argv = ['x']
import bench

dev = torch.device('cuda:0')
P = 64
X, y, bt = bench.synth_logit(D, dev, rows, P)
nn = torch.ones(rows, dtype=torch.float64, device=dev)
sh = D.GibbsShard(X, y, nn, seed=20240004)
sh.set_prior(np.zeros(P), np.eye(P) * 0.01)
sh.set_bp_local()
sh.finish_bp()
sh.set_beta(np.zeros(P))
ev = lambda: torch.cuda.Event(enable_timing=True)
ts, tb, tt = [], [], []
for s in range(60):
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    sh.sweep_local(s, None)
    e1.record()
    sh.draw_beta(s, 1)
    e2.record()
    e2.synchronize()
    ts.append(e0.elapsed_time(e1)); tb.append(e1.elapsed_time(e2)); tt.append(e0.elapsed_time(e2))
D.sync_status()
f = lambda v: f"median {np.median(v[10:]):.4f} min {min(v[10:]):.4f}"
print(f"rows {rows}: X pass ms {f(ts)} | beta stage ms {f(tb)} | both ms {f(tt)} -> {1e3 / np.median(tt[10:]):.0f} sweeps/s before the exchange")
# the same, untimed inside: what the stream sustains
torch.cuda.synchronize()
e0, e1 = ev(), ev()
e0.record()
for s in range(60, 160):
    sh.sweep_local(s, None)
    sh.draw_beta(s, 1)
e1.record()
e1.synchronize()
print(f"rows {rows}: 100 sweeps back to back: {e0.elapsed_time(e1) / 100:.4f} ms per sweep")
D.sync_status()
