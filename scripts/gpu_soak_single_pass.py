"""Randomised soak of the single-pass sweeps (P = 64: kernels_sweep1.hip, P = 256: kernels_sweep256.hip) against the two streaming
passes on the GPU: random row counts (1 .. 3e5; tails of every length), coefficient scales from 0 to 30 (nearly no to nearly all
rows deferred), shapes n up to 4, with and without omega requested.  Same omega to 1e-13 (bit for bit on the rows the fast path
settles), X'Omega X to summation order, exactly symmetric.
    python scripts/gpu_soak_single_pass.py [cases]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bayeslogit_amd import device as D

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(31337)
tot_def = 0
for c in range(cases):
    P = 64 if c % 2 == 0 else 256
    N = int(rng.choice([1, 3, 15, 16, 17, 31, 33, 255, 1000, 4097, 20011, 65537, 100003, 300007 if P == 64 else 120011]))
    scale = float(rng.choice([0.0, 0.3, 1.0, 3.0, 8.0, 30.0]))
    nmax = int(rng.choice([1, 1, 1, 2, 4]))
    X = torch.as_tensor(rng.standard_normal((N, P)) / np.sqrt(P), dtype=torch.float64, device=dev)
    n = torch.as_tensor(rng.integers(1, nmax + 1, N).astype(float), device=dev)
    y = torch.zeros(N, dtype=torch.float64, device=dev)
    beta0 = rng.standard_normal(P) * scale
    seed, idx0, sweep = int(rng.integers(1, 2**40)), int(rng.integers(0, 2**40)), int(rng.integers(0, 5000))
    out = {}
    for mode in (0, 1):
        D.set_sweep_mode(bool(mode))
        g = D.GibbsShard(X, y, n, seed=seed, idx0=idx0)
        g.set_beta(beta0)
        w = torch.full((N,), -1.0, dtype=torch.float64, device=dev)
        D.sweep_deferred_rows()
        g.sweep_local(sweep, w)
        D.sync_status()
        nd = D.sweep_deferred_rows()
        PP = g.pp().cpu().numpy().reshape(P, P).copy()
        g.sweep_local(sweep, None)
        D.sync_status()
        assert np.array_equal(PP, g.pp().cpu().numpy().reshape(P, P)), (c, "omega requested or not: different X'Omega X")
        out[mode] = (w.cpu().numpy(), PP, nd)
        g.close()
    D.set_sweep_mode(True)
    (w0, PP0, _), (w1, PP1, nd) = out[0], out[1]
    tot_def += nd
    assert np.all(w1 > 0) and np.allclose(w1, w0, rtol=1e-13, atol=0), (c, P, N, scale)
    assert (w1 != w0).sum() <= nd, (c, "a settled row differs")
    assert np.array_equal(PP1, PP1.T) and np.abs(PP1 - PP0).max() <= 1e-13 * max(np.abs(PP0).max(), 1e-300), (c, P, N, scale)
print(f"{cases} cases (P = 64 and 256 alternating): omega and X'Omega X of the single pass equal the two passes'; {tot_def} rows went through the deferred kernels")
