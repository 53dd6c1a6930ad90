"""First GPU contact: parity of every sampler vs the oracle on small inputs + rough timings."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import oracle_lib as O
import bayeslogit_amd as bl
from bayeslogit_amd import device as D

dev = torch.device("cuda:0")
def cmp(name, a, b, tol=1e-10):
    a = np.asarray(a); b = np.asarray(b)
    rel = np.abs(a - b) / np.maximum(1e-300, np.maximum(np.abs(a), np.abs(b)))
    bad = (rel > tol).sum()
    print(f"{name}: n={a.size} max_rel={rel.max():.3e} mismatches(>{tol})={bad}", flush=True)

N = 200000
rng = np.random.default_rng(1)
z = rng.uniform(0, 4, N)
zt = torch.tensor(z, device=dev)
x = D.rpg_devroye(zt, 1, seed=11, epoch=3, idx0=5); D.sync_status()
cmp("devroye b=1", x.cpu().numpy(), O.rpg_devroye(N, 1, z, 11, 3, 5))
nv = rng.integers(0, 4, N).astype(np.int32)
x = D.rpg_devroye(zt, torch.tensor(nv, device=dev), seed=12); D.sync_status()
cmp("devroye n vec", x.cpu().numpy(), O.rpg_devroye(N, nv, z, 12))
M = 50000
zz = rng.normal(0, 1.4142, M); zzt = torch.tensor(zz, device=dev)
for nm, h in [("alt", rng.uniform(1, 13, M)), ("alt4", np.full(M, 4.0)), ("alt1", np.full(M, 1.0))]:
    ht = torch.tensor(h, device=dev)
    x = D.rpg_alt(ht, zzt, seed=13); D.sync_status()
    cmp("alt " + nm, x.cpu().numpy(), O.rpg_alt(M, h, zz, 13))
h = rng.integers(14, 171, M).astype(float); ht = torch.tensor(h, device=dev)
it = torch.zeros(M, dtype=torch.int32, device=dev)
x = D.rpg_sp(ht, zzt, seed=14, iters=it); D.sync_status()
xo, ito = O.rpg_sp(M, h, zz, 14)
cmp("sp", x.cpu().numpy(), xo); print(" sp iter equal:", (it.cpu().numpy() == ito).mean())
h = rng.uniform(0.05, 0.99, 2000); ht = torch.tensor(h, device=dev); zs = torch.tensor(zz[:2000], device=dev)
x = D.rpg_gamma(ht, zs, 200, seed=15); D.sync_status()
cmp("gamma", x.cpu().numpy(), O.rpg_gamma(2000, h, zz[:2000], 15), 1e-9)
h = np.concatenate([rng.integers(1, 51, M - 3000).astype(float), rng.uniform(0.1, 200, 3000)]); rng.shuffle(h)
ht = torch.tensor(h, device=dev)
x = D.rpg_hybrid(ht, zzt, seed=16); D.sync_status()
cmp("hybrid", x.cpu().numpy(), O.rpg_hybrid(M, h, zz, 16), 1e-9)

# .C boundary
bl.set_seed(77)
x = bl.rpg_devroye(1000, 1, 0.0)
print("rpg.devroye .C mean/var", x.mean(), x.var(), "vs", 0.25, 1/24)
cmp(".C rpg_devroye", x, O.rpg_devroye(1000, 1, 0.0, 77, 0, 0))

# timings C2
for n in [10_000_000, 100_000_000]:
    zt = torch.empty(n, dtype=torch.float64, device=dev); D.fill_unif(zt, 0.0, 4.0, 20240001)
    out = torch.empty_like(zt)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        D.rpg_devroye(zt, 1, seed=20240002, out=out); torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"C2 N={n}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} M draws/s", flush=True)
    print(" mean", out.mean().item())
    del zt, out
n = 100_000_000
zt = torch.empty(n, dtype=torch.float64, device=dev); D.fill_norm(zt, 0.0, 2 ** 0.5, 20240001)
ht = torch.empty(n, dtype=torch.float64, device=dev); D.fill_shape(ht, 50, 20240001, epoch=1)
out = torch.empty_like(zt)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    D.rpg_hybrid(ht, zt, seed=20240002, out=out); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"C3 N={n}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} M draws/s", flush=True)
D.sync_status()
del zt, ht, out

# gibbs small parity
for (Nn, P) in [(1000, 64), (777, 16), (500, 10), (300, 48)]:
    X = rng.normal(size=(Nn, P)) / np.sqrt(P); X[:, -1] = 1.0
    bt = np.abs(rng.normal(size=P)); bt[-1] = -0.5
    y = (rng.uniform(size=Nn) < 1 / (1 + np.exp(-X @ bt))).astype(float)
    nn = np.ones(Nn); m0 = np.zeros(P); P0 = np.eye(P) * 0.01
    Xt = torch.tensor(X, device=dev); yt = torch.tensor(y, device=dev); nt = torch.tensor(nn, device=dev)
    g = D.GibbsShard(Xt, yt, nt, seed=99)
    g.set_prior(m0, P0); g.set_bp_local(); g.finish_bp(); g.set_beta(np.zeros(P))
    wt = torch.zeros(Nn, dtype=torch.float64, device=dev)
    g.sweep_local(0, wt); torch.cuda.synchronize()
    PPo, wo = O.sweep_partial(X, nn, np.zeros(P), 99, 0)
    cmp(f"gibbs N={Nn} P={P} omega", wt.cpu().numpy(), wo)
    PP = g.pp().cpu().numpy().reshape(P, P)
    print("  PP rel err", np.abs(PP - PPo).max() / np.abs(PPo).max(), "sym", np.abs(PP - PP.T).max())
    bPo = O.set_bP(y, X, nn, m0, P0)
    print("  bP err", np.abs(g.bp().cpu().numpy() - bPo).max())
    for con in (0, 1):
        g.set_beta(np.zeros(P)); g.sweep_local(0, None); g.draw_beta(0, con); D.sync_status()
        bo = O.draw_beta(PPo + P0, bPo, np.zeros(P), 99, 0, con)
        print(f"  beta constrain={con} max abs err", np.abs(g.get_beta() - bo).max())
    for con in (0, 1):
        t0 = time.time(); bg = g.run(20, 5, con); dt = time.time() - t0
        _, bo = O.gibbs(y, X, nn, m0, P0, 20, 5, 99, con, store_w=False)
        print(f"  chain constrain={con}: first-slot err {np.abs(bg[0]-bo[0]).max():.3e} last-slot err {np.abs(bg[-1]-bo[-1]).max():.3e} ({dt*1e3:.1f} ms)")
    g.close()
# EM
X = rng.normal(size=(2000, 8)); bt = rng.normal(size=8); y = (rng.uniform(size=2000) < 1 / (1 + np.exp(-X @ bt))).astype(float)
r = bl.logit_EM(y, X); bo, ito = O.em(y, X, np.ones(2000))
print("EM iters", r["iter"], ito, "beta err", np.abs(r["beta"] - bo).max())
# combine
Xd = rng.integers(0, 3, size=(5000, 3)).astype(float); yd = rng.uniform(size=5000); nd = rng.integers(1, 4, 5000).astype(float)
r = bl.logit_combine(yd, Xd, nd); yo, Xo, no = O.combine(yd, Xd, nd)
print("combine N", len(r["y"]), len(yo), "err", np.abs(r["y"] - yo).max(), np.abs(r["X"] - Xo).max(), np.abs(r["n"] - no).max())
# timing C4
Nn, P = 10_000_000, 64
Xt = torch.empty((Nn, P), dtype=torch.float64, device=dev); D.fill_norm(Xt, 0.0, 1 / 8.0, 20240003); Xt[:, -1] = 1.0
bt = torch.empty(P, dtype=torch.float64, device=dev); D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1); bt.abs_(); bt[-1] = -0.5
yt = torch.empty(Nn, dtype=torch.float64, device=dev); D.fill_logit_y(yt, Xt, bt, 20240003, epoch=2)
nt = torch.ones(Nn, dtype=torch.float64, device=dev)
g = D.GibbsShard(Xt, yt, nt, seed=20240004); g.set_prior(np.zeros(P), np.eye(P) * 0.01)
for con in (0, 1):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time(); b = g.run(20, 2, con); dt = time.time() - t0
        print(f"C4 constrain={con}: 22 sweeps {dt*1e3:.1f} ms -> {22/dt:.1f} sweeps/s; beta[:3]={b[-1][:3]}", flush=True)
print("DONE")
