"""The replicated P x P stage alone (PP + P0, Cholesky, beta draw) on posteriors of the shape a chain sees.
    python scripts/gpu_beta.py P [P ...]        (BL_BETA_SPLIT=0: the one-wavefront sweeps for 64 < P <= 256)
The digest printed per line covers every bit of the six draws' beta: equal digests across BL_BETA_SPLIT settings = same chain."""
import hashlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from bayeslogit_amd import device as D
from test_gpu_gibbs import _beta_problem, synth

dev = torch.device("cuda:0")
for P in [int(v) for v in sys.argv[1:]] or [64, 128, 256]:
    PPsum, P0, bP, bprev = _beta_problem(P, 100 + P)
    X, y, n = synth(8, P, 1)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    g = D.GibbsShard(t(X), t(y), t(n), seed=4321)
    g.set_prior(np.zeros(P), P0)
    ppd, bpd, bd = t(np.asfortranarray(PPsum).ravel(order="F")), t(bP), t(bprev)
    for con in (1, 0):
        ms = []
        dig = hashlib.sha1()
        for s in range(6):
            g.pp().copy_(ppd); g.bp().copy_(bpd); g.beta().copy_(bd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.draw_beta(s, con)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            dig.update(g.beta().cpu().numpy().tobytes())
        print(f"P={P} constrain={con}: {np.median(ms[1:]):.3f} ms  beta[:3]={g.beta()[:3].cpu().numpy()} digest={dig.hexdigest()[:12]}")
    D.sync_status()
    g.close()
