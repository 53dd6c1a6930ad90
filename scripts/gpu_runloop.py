"""C-side chain loop (bl_gibbs_run_stream) vs the Python driver loop: sweeps/s at C4 size."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
from bayeslogit_amd.dist import DistGibbs
dev = torch.device("cuda:0")
N, P = int(os.environ.get("BL_N", "10000000")), 64
X = torch.empty((N, P), dtype=torch.float64, device=dev); D.fill_norm(X, 0.0, 1 / 8.0, 20240003); X[:, -1] = 1.0
bt = torch.empty(P, dtype=torch.float64, device=dev); D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1); bt.abs_(); bt[-1] = -0.5
y = torch.empty(N, dtype=torch.float64, device=dev); D.fill_logit_y(y, X, bt, 20240003, epoch=2)
nn = torch.ones(N, dtype=torch.float64, device=dev)
g = D.GibbsShard(X, y, nn, seed=20240004); g.set_prior(np.zeros(P), np.eye(P) * 0.01)
for con in (1, 0):
    g.run_stream(20, 5, con, store_w="none")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = g.run_stream(300, 0, con, store_w="none", moments=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"C loop constrain={con}: {300/dt:.1f} sweeps/s ({dt/300*1e3:.3f} ms/sweep)", flush=True)
    drv = DistGibbs(g); drv.setup(np.zeros(P), np.eye(P) * 0.01, np.zeros(P))
    for s in range(20): drv.sweep(s, con)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(20, 320): drv.sweep(s, con)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"Python loop constrain={con}: {300/dt:.1f} sweeps/s ({dt/300*1e3:.3f} ms/sweep)", flush=True)
