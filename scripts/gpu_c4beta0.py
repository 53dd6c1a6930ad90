"""Sweep time with beta = 0 (psi = 0: no |psi| >= 3.125 rows) vs a posterior beta."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
def tm(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
N, P = 10_000_000, 64
X = torch.empty((N, P), dtype=torch.float64, device=dev); D.fill_norm(X, 0.0, 1 / 8.0, 20240003); X[:, -1] = 1.0
bt = torch.empty(P, dtype=torch.float64, device=dev); D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1); bt.abs_(); bt[-1] = -0.5
y = torch.empty(N, dtype=torch.float64, device=dev); D.fill_logit_y(y, X, bt, 20240003, epoch=2)
nn = torch.ones(N, dtype=torch.float64, device=dev)
g = D.GibbsShard(X, y, nn, seed=20240004); g.set_prior(np.zeros(P), np.eye(P) * 0.01); g.set_bp_local(); g.finish_bp()
sw = [10]
def sweep(): g.sweep_local(sw[0], None); sw[0] += 1
for name, b in (("beta=0", np.zeros(P)), ("beta=0.02", np.full(P, 0.02)), ("beta=true", bt.cpu().numpy()), ("beta=3*true", 3 * bt.cpu().numpy())):
    g.set_beta(b)
    psi = X @ torch.tensor(b, dtype=torch.float64, device=dev)
    frac = (psi.abs() >= 3.125).double().mean().item()
    print(f"{name}: sweep {tm(sweep):.3f} ms   frac |psi|>=3.125: {frac:.4f}", flush=True)
D.sync_status()
