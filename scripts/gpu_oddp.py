"""Sweep time for P that the MFMA paths do not take."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
def tm(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for (N, P) in [(2_000_000, 64), (2_000_000, 50), (2_000_000, 63), (1_000_000, 100), (1_000_000, 128), (500_000, 200), (500_000, 256)]:
    X = torch.empty((N, P), dtype=torch.float64, device=dev); D.fill_norm(X, 0.0, 1 / P ** 0.5, 20240003); X[:, -1] = 1.0
    y = (torch.rand(N, device=dev, dtype=torch.float64) < 0.5).double()
    nn = torch.ones(N, dtype=torch.float64, device=dev)
    g = D.GibbsShard(X, y, nn, seed=1); g.set_prior(np.zeros(P), np.eye(P) * 0.01); g.set_bp_local(); g.finish_bp()
    g.set_beta(np.full(P, 0.05))
    sw = [0]
    def sweep(): g.sweep_local(sw[0], None); sw[0] += 1
    ts = tm(sweep)
    print(f"N={N} P={P}: sweep {ts:.3f} ms  ({8*N*P/ts/1e6:.0f} GB/s of X)", flush=True)
    D.sync_status(); g.close(); del X, y, nn
