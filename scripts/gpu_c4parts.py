"""Parts of the C4 sweep timed separately: EM pass (psi only), draw on a psi vector, full sweep."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
N, P = 10_000_000, 64
X = torch.empty((N, P), dtype=torch.float64, device=dev); D.fill_norm(X, 0.0, 1 / 8.0, 20240003); X[:, -1] = 1.0
bt = torch.empty(P, dtype=torch.float64, device=dev); D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1); bt.abs_(); bt[-1] = -0.5
y = torch.empty(N, dtype=torch.float64, device=dev); D.fill_logit_y(y, X, bt, 20240003, epoch=2)
nn = torch.ones(N, dtype=torch.float64, device=dev)
g = D.GibbsShard(X, y, nn, seed=20240004); g.set_prior(np.zeros(P), np.eye(P) * 0.01); g.set_bp_local(); g.finish_bp()
g.set_beta(np.zeros(P))
for s in range(3):
    g.sweep_local(s, None); g.draw_beta(s, 0)
for s in range(5):
    g.sweep_local(10 + s, None)
for s in range(5):
    g.em_local()
beta = torch.tensor(g.get_beta(), dtype=torch.float64, device=dev)
psi = X @ beta
out = torch.empty_like(psi)
for s in range(5):
    D.rpg_devroye(psi, 1, seed=5, out=out)
D.sync_status()
print("psi mean/sd", psi.mean().item(), psi.std().item())
