import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bayeslogit_amd as bl
rng = np.random.default_rng(1)
N, P = 10_000_000, 64
X = rng.standard_normal((N, P)) / 8.0
X[:, -1] = 1.0
bt = np.abs(rng.standard_normal(P)); bt[-1] = -0.5
y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ bt))).astype(float)
import ctypes as C
from bayeslogit_amd import _lib
w = np.empty((3, N)); w.fill(0)
beta = np.zeros((3, P))
n = np.ones(N)
m0 = np.zeros(P); P0 = np.asfortranarray(np.eye(P) * 0.01)
dp = lambda a: a.ctypes.data_as(_lib.c_dp)
bl.set_seed(5)
for rep in range(2):
    t0 = time.perf_counter()
    _lib.lib().gibbs(dp(w), dp(beta), dp(y), dp(X), dp(n), dp(m0), dp(P0), C.byref(C.c_int(N)), C.byref(C.c_int(P)), C.byref(C.c_int(3)), C.byref(C.c_int(2)))
    dt = time.perf_counter() - t0
    print(f"gibbs .C N=1e7 P=64 burn 2 + samp 3: {dt:.3f} s (upload 5.3 GB, omega out 0.24 GB); beta[-1][:3] {beta[-1][:3]}, w mean {w.mean():.5f}")
