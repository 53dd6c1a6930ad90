import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
for (N, P) in [(1_000_000, 256), (2_000_000, 128)]:
    X = torch.empty((N, P), dtype=torch.float64, device=dev); D.fill_norm(X, 0.0, 1 / P ** 0.5, 20240003); X[:, -1] = 1.0
    bt = torch.empty(P, dtype=torch.float64, device=dev); D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1); bt.abs_(); bt[-1] = -0.5
    y = torch.empty(N, dtype=torch.float64, device=dev); D.fill_logit_y(y, X, bt, 20240003, epoch=2)
    nn = torch.ones(N, dtype=torch.float64, device=dev)
    g = D.GibbsShard(X, y, nn, seed=20240004); g.set_prior(np.zeros(P), np.eye(P) * 0.01); g.set_bp_local(); g.finish_bp()
    g.set_beta(np.zeros(P))
    def tm(fn, reps=3):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return float(np.median(ts))
    sw = [0]
    def sweep(): g.sweep_local(sw[0], None); sw[0] += 1
    ts = tm(sweep); tb0 = tm(lambda: g.draw_beta(1, 0)); 
    print(f"N={N} P={P}: sweep {ts:.2f} ms ({8*N*P/ts/1e6:.0f} GB/s alg, {N*P*P/ts/1e9:.2f} TFLOP/s sym)  beta(unconstrained) {tb0:.2f} ms", flush=True)
    tb1 = tm(lambda: g.draw_beta(1, 1), 2); print(f"   beta(constrained) {tb1:.2f} ms", flush=True)
    D.sync_status(); g.close(); del X, y, nn
