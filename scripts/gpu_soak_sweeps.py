"""Soak of the constrained coordinate sweeps for 64 < P <= 256: the row-split segment kernel (with its hand-over to the
one-wavefront kernel for pressed chains) against the one-wavefront kernel alone, bit for bit, over random widths, posteriors
from slack to pressed, and chains long enough to cross between the two regimes.
    python scripts/gpu_soak_sweeps.py [problems=40] [draws=16]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from bayeslogit_amd import device as D
from test_gpu_gibbs import synth

dev = torch.device("cuda:0")
nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ndraw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(20241005)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
bad = 0
for it in range(nprob):
    P = int(rng.integers(65, 257))
    rows_per_col = int(rng.choice([30, 60, 200, 800, 2500]))       # weak data: pressed; rich data: slack
    N = rows_per_col * P
    X = rng.normal(size=(N, P)) / np.sqrt(P)
    X[:, -1] = 1.0
    bt = np.abs(rng.normal(size=P)) * float(rng.choice([0.05, 0.3, 1.0])) + float(rng.choice([0.0, 0.0, 0.2]))
    bt[-1] = -0.5
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-(X @ bt)))).astype(np.float64)
    w = rng.gamma(2.0, 0.12, N)
    PP = (X * w[:, None]).T @ X
    P0 = np.eye(P) * float(rng.choice([0.01, 0.3]))
    bP = X.T @ (y - 0.5)
    bprev = np.abs(rng.normal(size=P)) * 0.3
    bprev[-1] = -0.4
    Xs, ys, ns = synth(8, P, 1)
    out = {}
    for kind in (1, 0):
        D.set_beta_sweeps(kind)
        g = D.GibbsShard(t(Xs), t(ys), t(ns), seed=977 + it)
        g.set_prior(np.zeros(P), P0)
        ppd, bpd = t(np.asfortranarray(PP).ravel(order="F")), t(bP)
        g.beta().copy_(t(bprev))
        draws = []
        for s in range(ndraw):
            # a posterior that tightens and loosens along the chain: the chain crosses between the regimes
            scale = 1.0 if (s // 4) % 2 == 0 else 0.02
            g.pp().copy_(ppd * scale)
            g.bp().copy_(bpd * scale)
            g.draw_beta(s, 1)
            draws.append(g.beta().cpu().numpy().copy())
        D.sync_status()
        g.close()
        out[kind] = np.stack(draws)
    same = np.array_equal(out[1], out[0])
    ok = same and np.all(np.isfinite(out[1])) and np.all(out[1][:, :-1] >= 0.0)
    bad += 0 if ok else 1
    print(f"problem {it:3d}: P = {P:3d}, N/P = {rows_per_col:4d}: {'same bits' if same else 'DIFFERENT'}"
          f"{'' if ok else '  <-- FAILED'}", flush=True)
D.set_beta_sweeps(1)
print("soak:", "all equal" if bad == 0 else f"{bad} problems differ")
sys.exit(1 if bad else 0)
