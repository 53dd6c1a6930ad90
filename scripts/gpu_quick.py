"""Quick timings: C2 draws, C3 mixed, C4 sweep + beta kernels."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
def tm(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), float(np.median(ts))
n = 100_000_000
z = torch.empty(n, dtype=torch.float64, device=dev); D.fill_unif(z, 0.0, 4.0, 20240001)
x = torch.empty_like(z)
mn, md = tm(lambda: D.rpg_devroye(z, 1, seed=20240002, out=x))
print(f"C2 devroye 1e8: min {mn:.2f} ms median {md:.2f} ms -> {n/md/1e3:.0f} M draws/s  mean={x.mean().item():.6f}", flush=True)
z0 = torch.zeros(n, dtype=torch.float64, device=dev)
mn, md = tm(lambda: D.rpg_devroye(z0, 1, seed=20240002, out=x))
print(f"   z=0      : median {md:.2f} ms -> {n/md/1e3:.0f} M draws/s", flush=True)
del z0
if "--mixed" in sys.argv:
    h = torch.empty(n, dtype=torch.float64, device=dev); D.fill_shape(h, 50, 20240001, epoch=1)
    zz = torch.empty(n, dtype=torch.float64, device=dev); D.fill_norm(zz, 0.0, 2 ** 0.5, 20240001)
    mn, md = tm(lambda: D.rpg_hybrid(h, zz, seed=20240002, out=x), reps=2)
    print(f"C3 hybrid 1e8: median {md:.2f} ms -> {n/md/1e3:.0f} M draws/s", flush=True)
    del h, zz
D.sync_status()
del z, x
N, P = 10_000_000, 64
X = torch.empty((N, P), dtype=torch.float64, device=dev); D.fill_norm(X, 0.0, 1 / 8.0, 20240003); X[:, -1] = 1.0
bt = torch.empty(P, dtype=torch.float64, device=dev); D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1); bt.abs_(); bt[-1] = -0.5
y = torch.empty(N, dtype=torch.float64, device=dev); D.fill_logit_y(y, X, bt, 20240003, epoch=2)
nn = torch.ones(N, dtype=torch.float64, device=dev)
g = D.GibbsShard(X, y, nn, seed=20240004); g.set_prior(np.zeros(P), np.eye(P) * 0.01); g.set_bp_local(); g.finish_bp()
g.set_beta(np.zeros(P))
for _ in range(3):
    g.sweep_local(0, None); g.draw_beta(0, 0)
sw = [1]
def sweep(): g.sweep_local(sw[0], None); sw[0] += 1
mn, md = tm(sweep, 8); print(f"C4 sweep kernel: median {md:.3f} ms ({8*N*P/md/1e6:.0f} GB/s)", flush=True)
for con in (0, 1):
    mn, md = tm(lambda: g.draw_beta(sw[0], con), 8); print(f"C4 beta draw constrain={con}: median {md:.3f} ms", flush=True)
D.sync_status()
print("beta head", g.get_beta()[:4])
