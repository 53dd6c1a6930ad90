"""Soak of the constrained beta draw (k_beta64's speculative groups, k_beta_sweeps for P > 64) against the
oracle's move-by-move draw: random P, N, seeds and beta_prev placements; reports the worst difference over the
cases where the oracle's own draw is stable under a 1e-15 perturbation of PP (tests/test_gpu_gibbs.py explains)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import oracle_lib as oracle
from bayeslogit_amd import device as D

oracle.build()
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("BL_SOAK_SEED", "1")))
ncase = int(os.environ.get("BL_SOAK_CASES", "300"))
worst, nstable, nun = 0.0, 0, 0
for case in range(ncase):
    P = int(rng.integers(1, 97)) if case % 5 else int(rng.choice([64, 63, 65, 32, 128]))
    N = int(P * rng.integers(20, 60))
    X = rng.normal(size=(N, P)) / np.sqrt(P); X[:, -1] = 1.0
    kind = case % 3
    bt = np.abs(rng.normal(size=P)) * (1.0 if kind == 0 else 0.3 if kind == 1 else 0.02); bt[-1] = -0.5
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ bt))).astype(float)
    n = np.ones(N)
    seed = int(rng.integers(1, 2**31))
    g = D.GibbsShard(torch.tensor(X, device=dev), torch.tensor(y, device=dev), torch.tensor(n, device=dev), seed=seed, idx0=3)
    m0, P0 = np.zeros(P), np.eye(P) * 0.2
    g.set_prior(m0, P0); g.set_bp_local(); g.finish_bp()
    bPo = oracle.set_bP(y, X, n, m0, P0)
    beta = bt.copy(); beta[:-1] = np.maximum(beta[:-1], 0.0)
    for sweep in range(2):
        g.set_beta(beta); g.sweep_local(sweep, None); g.draw_beta(sweep, 1); D.sync_status()
        PPo, _ = oracle.sweep_partial(X, n, beta, seed, sweep, 3)
        bo = oracle.draw_beta(PPo + P0, bPo, beta, seed, sweep, 1)
        E = rng.normal(size=(P, P)) * 1e-15
        bo2 = oracle.draw_beta((PPo + P0) * (1 + (E + E.T) / 2), bPo, beta, seed, sweep, 1)
        bg = g.get_beta()
        assert np.all(bg[:-1] >= -1e-12) and np.all(np.isfinite(bg)), (case, P)
        if np.abs(bo2 - bo).max() < 1e-10:
            nstable += 1
            dlt = np.abs(bg - bo).max()
            worst = max(worst, dlt)
            assert dlt < 1e-9, (case, P, N, kind, sweep, dlt)
        else:
            nun += 1
        beta = bo
    g.close()
    if case % 50 == 49:
        print(f"{case + 1} cases: stable draws {nstable}, unstable {nun}, worst |gpu - oracle| {worst:.3e}", flush=True)
print(f"done: stable draws {nstable}, unstable {nun}, worst |gpu - oracle| {worst:.3e}")
