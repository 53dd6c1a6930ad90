"""Oracle Gibbs / EM / combine / mlogit against numpy linear algebra, a plain-python restatement of the
merge, and the reference's own checks (posterior mean next to the EM mode, test_logit.cpp:69-74;
colMeans(beta) agreement, LogitTest.R:26-34,83-86)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


def synth(N, P, seed, nonneg=True):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, P)) / np.sqrt(P)
    X[:, -1] = 1.0
    bt = np.abs(rng.normal(size=P)) if nonneg else rng.normal(size=P)
    bt[-1] = -0.5
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ bt))).astype(float)
    return X, y, np.ones(N), bt


def irls(y, X, n, iters=50):
    b = np.zeros(X.shape[1])
    for _ in range(iters):
        p = 1 / (1 + np.exp(-X @ b))
        W = n * p * (1 - p)
        b = b + np.linalg.solve(X.T @ (X * W[:, None]), X.T @ (n * (y - p)))
    return b


def test_cholesky(oracle):
    rng = np.random.default_rng(0)
    for P in (1, 3, 17, 64):
        A = rng.normal(size=(P + 5, P))
        A = A.T @ A + 0.1 * np.eye(P)
        U = np.zeros((P, P), order="F")
        Lw = np.zeros((P, P), order="F")
        Af = np.asfortranarray(A)
        assert oracle.lib().bl_chol_upper(U.ctypes.data_as(O.c_dp), Af.ctypes.data_as(O.c_dp), P) == 0
        assert oracle.lib().bl_chol_lower(Lw.ctypes.data_as(O.c_dp), Af.ctypes.data_as(O.c_dp), P) == 0
        assert np.allclose(U.T @ U, A, rtol=1e-12) and np.allclose(np.triu(U), U)
        assert np.allclose(Lw @ Lw.T, A, rtol=1e-12) and np.allclose(np.tril(Lw), Lw)
    bad = np.asfortranarray(np.array([[1.0, 2.0], [2.0, 1.0]]))
    U = np.zeros((2, 2), order="F")
    assert oracle.lib().bl_chol_upper(U.ctypes.data_as(O.c_dp), bad.ctypes.data_as(O.c_dp), 2) == 2


def test_set_bP_and_sweep_partial(oracle):
    X, y, n, _ = synth(300, 7, 1)
    n = n * np.arange(1, 301) % 4 + 1.0
    m0 = np.arange(7) * 0.1
    P0 = np.eye(7) * 0.5 + 0.05
    bP = oracle.set_bP(y, X, n, m0, P0)
    assert np.allclose(bP, P0 @ m0 + X.T @ (n * (y - 0.5)), rtol=1e-12)      # Logit.hpp:174-190
    beta = np.linspace(-0.3, 0.4, 7)
    PP, w = oracle.sweep_partial(X, n, beta, seed=5, sweep=3, idx0=10)
    assert np.allclose(PP, X.T @ (X * w[:, None]), rtol=1e-12) and np.array_equal(PP, PP.T)
    # omega_i is PG((int) n_i, x_i.beta) on stream (seed, idx0+i, DOM_OMEGA = 3, sweep)
    import ctypes as C
    psi = np.array([sum(X[i, j] * beta[j] for j in range(7)) for i in range(300)])   # the oracle's summation order
    ref = []
    for i in range(300):
        r = oracle.rng(5, 10 + i, 3, 3)
        ref.append(oracle.lib().bl_pg_draw_devroye(int(n[i]), psi[i], C.byref(r)))
    assert np.allclose(w, ref, rtol=1e-12, atol=0)
    # ... and is not what an rpg_devroye call of the same (seed, epoch) would read (DOM_DRAW)
    assert not np.any(w == oracle.rpg_devroye(300, n.astype(np.int32), psi, 5, 3, 10))


def test_unconstrained_beta_is_mvn(oracle):
    """Logit.hpp:291-320: beta ~ N(PP^-1 bP, PP^-1)."""
    rng = np.random.default_rng(3)
    P = 5
    A = rng.normal(size=(20, P))
    PP = A.T @ A + np.eye(P)
    bP = rng.normal(size=P)
    draws = np.array([oracle.draw_beta(PP, bP, np.zeros(P), seed=8, sweep=s, constrain=0) for s in range(6000)])
    S = np.linalg.inv(PP)
    se = np.sqrt(np.diag(S) / 6000)
    assert np.all(np.abs(draws.mean(0) - S @ bP) < 5 * se)
    assert np.allclose(np.cov(draws.T), S, atol=6 * S.max() / np.sqrt(6000))


def test_constrained_beta_respects_bounds(oracle):
    """Logit.hpp:383-391: beta_j >= 0 for j < P-1, last coefficient free."""
    rng = np.random.default_rng(4)
    P = 6
    A = rng.normal(size=(30, P))
    PP = A.T @ A + np.eye(P)
    bP = rng.normal(size=P) - 2.0          # pushes the unconstrained mean negative
    prev = np.zeros(P)
    neg_last = 0
    for s in range(200):
        prev = oracle.draw_beta(PP, bP, prev, seed=9, sweep=s, constrain=1)
        assert np.all(prev[:-1] >= -1e-12)
        neg_last += prev[-1] < 0
    assert neg_last > 0


def test_constrained_matches_truncated_mvn_moments(oracle):
    """P=2: beta_0 >= 0, beta_1 free; compare the chain's moments with numerical integration."""
    PP = np.array([[2.0, 0.6], [0.6, 1.0]])
    bP = np.array([-0.5, 0.3])
    S = np.linalg.inv(PP)
    m = S @ bP
    g0 = np.linspace(0, 8, 4001)
    # marginal of beta_0 is N(m0, S00) truncated to >= 0; E[beta_1 | beta_0] linear
    from scipy import stats
    d = stats.truncnorm((0 - m[0]) / np.sqrt(S[0, 0]), np.inf, loc=m[0], scale=np.sqrt(S[0, 0]))
    e0 = d.mean()
    e1 = m[1] + S[1, 0] / S[0, 0] * (e0 - m[0])
    prev = np.zeros(2)
    acc = []
    for s in range(20000):
        prev = oracle.draw_beta(PP, bP, prev, seed=10, sweep=s, constrain=1)
        acc.append(prev)
    acc = np.array(acc)[500:]
    assert abs(acc[:, 0].mean() - e0) < 0.03 and abs(acc[:, 1].mean() - e1) < 0.03
    assert abs(acc[:, 0].var() - d.var()) < 0.03


def test_gibbs_slot_semantics_and_restart(oracle):
    X, y, n, _ = synth(200, 4, 2)
    m0, P0 = np.zeros(4), np.eye(4) * 0.1
    w, beta = oracle.gibbs(y, X, n, m0, P0, samp=6, burn=3, seed=11, constrain=0)
    assert w.shape == (6, 200) and beta.shape == (6, 4)
    # sweep s uses epoch s: re-deriving slot k by hand from slot k-1 reproduces it (Logit.hpp:426-444)
    bP = oracle.set_bP(y, X, n, m0, P0)
    for k in range(1, 6):
        PP, wk = oracle.sweep_partial(X, n, beta[k - 1], seed=11, sweep=3 + k)
        assert np.array_equal(wk, w[k])
        bk = oracle.draw_beta(PP + P0, bP, beta[k - 1], seed=11, sweep=3 + k, constrain=0)
        assert np.allclose(bk, beta[k], rtol=1e-12, atol=1e-14)
    # burn = 0 is legal; omega not stored gives the same beta
    _, b2 = oracle.gibbs(y, X, n, m0, P0, samp=6, burn=3, seed=11, constrain=0, store_w=False)
    assert np.array_equal(beta, b2)
    _, b0 = oracle.gibbs(y, X, n, m0, P0, samp=2, burn=0, seed=11, constrain=0)
    assert np.all(np.isfinite(b0))


@pytest.mark.parametrize("constrain", [0, 1])
def test_posterior_mean_near_mode(oracle, constrain):
    """test_logit.cpp:40-74: posterior mean of beta next to the EM mode (P=1, N=100, beta=0.5),
    plus a P=4 case with non-negative truth (so the fork's constraint is nearly inactive)."""
    rng = np.random.default_rng(5)
    X = rng.normal(size=(100, 1))
    y = (rng.uniform(size=100) < 1 / (1 + np.exp(-0.5 * X[:, 0]))).astype(float)
    n = np.ones(100)
    _, beta = oracle.gibbs(y, X, n, np.zeros(1), np.zeros((1, 1)), samp=3000, burn=100, seed=12, constrain=0)
    mode, it = oracle.em(y, X, n)
    assert it < 100 and abs(beta.mean() - mode[0]) < 0.15 * max(0.2, beta.std() * 3)
    X, y, n, bt = synth(1500, 4, 6)
    _, beta = oracle.gibbs(y, X, n, np.zeros(4), np.eye(4) * 0.01, samp=1500, burn=200, seed=13,
                           constrain=constrain)
    mle = irls(y, X, n)
    sd = beta.std(0)
    assert np.all(np.abs(beta.mean(0) - mle) < 0.6 * sd + 0.05)


def test_em_matches_irls(oracle):
    X, y, n, _ = synth(800, 6, 7, nonneg=False)
    n = np.random.default_rng(1).integers(1, 5, 800).astype(float)
    y = np.random.default_rng(2).binomial(n.astype(int), 1 / (1 + np.exp(-X @ np.linspace(-1, 1, 6)))) / n
    b, it = oracle.em(y, X, n, tol=1e-10, max_iter=500)
    assert 1 < it < 500
    assert np.allclose(b, irls(y, X, n), atol=1e-6)
    b1, it1 = oracle.em(y, X, n, tol=1e-9, max_iter=3)       # max_iter honoured, returns iteration count
    assert it1 == 3


def py_combine(y, X, n):
    """Plain restatement of Logit::compress (Logit.hpp:192-270) for cross-checking."""
    y, X, n = list(y), [tuple(r) for r in X], list(n)
    i = 0
    while i < len(X):
        j = i + 1
        while j < len(X):
            if X[i] == X[j]:
                s = n[i] + n[j]
                y[i] = (n[i] / s) * y[i] + (n[j] / s) * y[j]
                n[i] = s
                del X[j], y[j], n[j]
            else:
                j += 1
        i += 1
    return np.array(y), np.array(X), np.array(n)


def test_combine(oracle):
    rng = np.random.default_rng(8)
    X = rng.integers(0, 3, size=(400, 3)).astype(float)
    X[5] = X[17]
    y = rng.uniform(size=400)
    n = rng.integers(1, 4, 400).astype(float)
    yo, Xo, no = oracle.combine(y, X, n)
    yp, Xp, npp = py_combine(y, X, n)
    assert np.array_equal(Xo, Xp) and np.array_equal(no, npp) and np.array_equal(yo, yp)
    assert no.sum() == n.sum() and np.isclose((yo * no).sum(), (y * n).sum())
    # nothing to merge: identity
    Xu = rng.normal(size=(50, 2))
    yu, Xuo, nu = oracle.combine(y[:50], Xu, n[:50])
    assert np.array_equal(Xuo, Xu) and np.array_equal(yu, y[:50])
    # -0.0 == 0.0 merges; NaN never equals itself
    Xz = np.array([[0.0, 1.0], [-0.0, 1.0], [np.nan, 1.0], [np.nan, 1.0]])
    yz, Xzo, nz = oracle.combine(np.array([0.0, 1.0, 0.5, 0.5]), Xz, np.ones(4))
    assert len(yz) == 3 and nz[0] == 2 and yz[0] == 0.5


def test_mult_combine_and_gibbs(oracle):
    rng = np.random.default_rng(9)
    N, P, J = 600, 3, 3
    X = rng.normal(size=(N, P))
    X[:, -1] = 1.0
    B = np.array([[1.0, -0.5], [0.0, 0.8], [0.2, -0.2]])
    eta = np.concatenate([X @ B, np.zeros((N, 1))], axis=1)
    pr = np.exp(eta) / np.exp(eta).sum(1, keepdims=True)
    cat = np.array([rng.choice(J, p=p) for p in pr])
    y = np.zeros((N, J - 1))
    for j in range(J - 1):
        y[cat == j, j] = 1.0
    n = np.ones(N)
    yc, Xc, nc = oracle.mult_combine(y, X, n)
    assert len(nc) == N
    Xd = np.repeat(X[:5], 2, axis=0)
    yd = np.repeat(y[:5], 2, axis=0)
    yc, Xc, nc = oracle.mult_combine(yd, Xd, np.ones(10))
    assert len(nc) == 5 and np.all(nc == 2) and np.array_equal(Xc, X[:5]) and np.array_equal(yc, y[:5])
    w, beta = oracle.mult_gibbs(y, X, n, np.zeros((P, J - 1)), np.zeros((P, P, J - 1)) + np.eye(P)[:, :, None] * 0.01,
                                samp=600, burn=100, seed=14)
    assert w.shape == (600, N, J - 1) and beta.shape == (600, P, J - 1)
    est = beta[100:].mean(0)
    assert np.abs(est - B).max() < 0.45
    # J = 2 reduces to the binomial model: same posterior as logit() (unconstrained draw)
    y2 = (cat == 0).astype(float)[:, None]
    _, b2 = oracle.mult_gibbs(y2, X, n, np.zeros((P, 1)), np.eye(P)[:, :, None] * 0.01, 800, 100, seed=15)
    _, b1 = oracle.gibbs(y2[:, 0], X, n, np.zeros(P), np.eye(P) * 0.01, 800, 100, seed=16, constrain=0)
    assert np.all(np.abs(b2[:, :, 0].mean(0) - b1.mean(0)) < 4 * b1.std(0) / np.sqrt(800 / 10) + 0.02)
