"""Phase timing of the constrained beta stage (BL_BETA_DEBUG=1) on a C4-shaped problem of 1e6 rows."""
import sys, os
os.environ["BL_BETA_DEBUG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
N, P = int(os.environ.get("BL_N", "1000000")), int(os.environ.get("BL_P", "64"))
X = torch.empty((N, P), dtype=torch.float64, device=dev); D.fill_norm(X, 0.0, 1 / P ** 0.5, 20240003); X[:, -1] = 1.0
bt = torch.empty(P, dtype=torch.float64, device=dev); D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1); bt.abs_(); bt[-1] = -0.5
y = torch.empty(N, dtype=torch.float64, device=dev); D.fill_logit_y(y, X, bt, 20240003, epoch=2)
nn = torch.ones(N, dtype=torch.float64, device=dev)
g = D.GibbsShard(X, y, nn, seed=20240004); g.set_prior(np.zeros(P), np.eye(P) * 0.01); g.set_bp_local(); g.finish_bp()
g.set_beta(np.zeros(P))
for s in range(30):
    g.sweep_local(s, None); g.draw_beta(s, 1)
D.sync_status()
