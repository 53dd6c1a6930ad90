"""mlogit through the .C boundary: wall time per sweep (includes H2D of X and D2H of w)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import bayeslogit_amd as bl
rng = np.random.default_rng(0)
N, P, J = 1_000_000, 32, 4
X = rng.normal(size=(N, P)) / np.sqrt(P); X[:, -1] = 1.0
B = rng.normal(size=(P, J - 1))
eta = np.concatenate([X @ B, np.zeros((N, 1))], axis=1)
pr = np.exp(eta - eta.max(1, keepdims=True)); pr /= pr.sum(1, keepdims=True)
cat = (rng.random(N)[:, None] > pr.cumsum(1)).sum(1)
y = np.zeros((N, J - 1)); 
for j in range(J - 1): y[:, j] = (cat == j)
m0 = np.zeros((P, J - 1)); P0 = np.zeros((P, P, J - 1))
for j in range(J - 1): P0[:, :, j] = np.eye(P) * 0.01
bl.set_seed(1)
for samp, burn in ((2, 1), (20, 10), (40, 20)):
    t0 = time.perf_counter()
    out = bl.mlogit(y, X, None, m0, P0, samp=samp, burn=burn)
    dt = time.perf_counter() - t0
    print(f"samp={samp} burn={burn}: {dt:.3f} s  -> {(dt)/(samp+burn)*1e3:.2f} ms/sweep incl. transfers; beta[-1,:3,0]={out['beta'][-1,:3,0]}", flush=True)
