"""The single-pass sweep (kernels_sweep1.hip) next to the two-pass sweep on the C4-shaped problem: omega bit for bit,
X' Omega X to summation order, time per sweep, rows that left the fast path.   BL_N=10000000 python scripts/gpu_sweep1.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bayeslogit_amd import device as D, _lib
import ctypes as C
sys.argv = ['x']
import bench
dev = torch.device('cuda:0')
N, P = int(os.environ.get('BL_N', '10000000')), 64
L = _lib.lib()
X, y, bt = bench.synth_logit(D, dev, N, P)
nn = torch.ones(N, dtype=torch.float64, device=dev)
sh = D.GibbsShard(X, y, nn, seed=20240004)
sh.set_prior(np.zeros(P), np.eye(P) * 0.01); sh.set_bp_local(); sh.finish_bp()


def deferred():
    v = C.c_uint64(0)
    _lib.check(L.bl_diag_sweep_deferred(C.byref(v)), "diag")
    return v.value


cases = (("beta = 0", np.zeros(P)), ("beta = beta_true", bt.cpu().numpy()), ("beta = 3 beta_true", 3 * bt.cpu().numpy()))
for name, beta in cases[:int(os.environ.get('BL_CASES', '3'))]:
    sh.set_beta(beta)
    res = {}
    for mode in (0, 1):
        L.bl_set_sweep_mode(mode)
        w = torch.zeros(N, dtype=torch.float64, device=dev)
        deferred()
        sh.sweep_local(7, w)
        torch.cuda.synchronize()
        nd = deferred()
        pp = sh.pp().clone()
        # timing: omega not stored
        for _ in range(3):
            sh.sweep_local(8, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in range(20):
            sh.sweep_local(9 + s, None)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = (w, pp, e0.elapsed_time(e1) / 20, nd)
    D.sync_status()
    w0, pp0, t0, _ = res[0]
    w1, pp1, t1, nd = res[1]
    same = bool(torch.equal(w0, w1))
    nbad = int((w0 != w1).sum())
    rel = float(((pp0 - pp1).abs().max() / pp0.abs().max()))
    sym = float((pp1.view(P, P) - pp1.view(P, P).t()).abs().max())
    if nbad:
        ii = torch.nonzero(w0 != w1).flatten()[:8]
        bt_ = torch.as_tensor(beta, device=dev)
        for i in ii.tolist():
            print(f"   row {i}: two-pass {float(w0[i])!r} single-pass {float(w1[i])!r} psi {float(X[i] @ bt_)!r}", flush=True)
    print(f"{name}: omega identical {same} ({nbad} differ), PP rel diff {rel:.2e}, asym {sym:.1e}, "
          f"two-pass {t0:.3f} ms, single-pass {t1:.3f} ms, deferred rows {nd} ({100.0 * nd / N:.2f} %)", flush=True)
L.bl_set_sweep_mode(1)
