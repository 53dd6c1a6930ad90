"""R-level API mirror (Code/R/LogitWrapper.R), calling the `.C` boundary.

Each function restates the parameter checks, recycling rules and output
reshaping of the R function it mirrors (cited per function) and hands caller-owned
host buffers to the same-named C symbol, exactly as `.C(name, ...)` does.
Parameter-check failures print the reference's message and return None / -1 in
place of R's NA / -1.
"""
import ctypes as C

import numpy as np

from . import _lib

_NA = None


def set_seed(seed):
    """Seed of the counter RNG behind the .C entry points (R: set.seed)."""
    _lib.lib().bl_set_seed(int(seed) & (2**64 - 1))


def _dp(a):
    return a.ctypes.data_as(_lib.c_dp)


def _ip(a):
    return a.ctypes.data_as(_lib.c_ip)


def _recycle(v, num, dtype=np.float64):
    """R's `array(v, num)`: recycle to length num."""
    v = np.atleast_1d(np.asarray(v, dtype=dtype)).ravel()
    if v.size != num:
        v = np.resize(v, num)
    return np.ascontiguousarray(v)


def _ci(v):
    return C.byref(C.c_int(int(v)))


def rpg_gamma(num=1, h=1, z=0.0, trunc=200):
    """LogitWrapper.R:12-32."""
    h_a = np.atleast_1d(np.asarray(h, dtype=np.float64))
    if np.sum(h_a < 0) != 0:
        print("h must be greater than zero.")
        return _NA
    if trunc < 1:
        print("trunc must be > 0.")
        return _NA
    _lib.require_gpu()
    x = np.zeros(num)
    h_a = _recycle(h, num)
    z_a = _recycle(z, num)
    _lib.lib().rpg_gamma(_dp(x), _dp(h_a), _dp(z_a), _ci(num), _ci(trunc))
    return x


def rpg_devroye(num=1, n=1, z=0.0):
    """LogitWrapper.R:34-51."""
    if np.any(np.asarray(n) < 0):
        print("n must be greater than zero.")
        return _NA
    _lib.require_gpu()
    x = np.zeros(num)
    n_a = _recycle(n, num, np.int32)
    z_a = _recycle(z, num)
    _lib.lib().rpg_devroye(_dp(x), _ip(n_a), _dp(z_a), _ci(num))
    return x


def rpg_alt(num=1, h=1, z=0.0):
    """LogitWrapper.R:53-71."""
    if np.any(np.asarray(h) < 1):
        print("h must be >= 1.")
        return _NA
    _lib.require_gpu()
    x = np.zeros(num)
    h_a = _recycle(h, num)
    z_a = _recycle(z, num)
    _lib.lib().rpg_alt(_dp(x), _dp(h_a), _dp(z_a), _ci(num))
    return x


def rpg_sp(num=1, h=1, z=0.0, track_iter=False):
    """LogitWrapper.R:74-100."""
    if np.any(np.asarray(h) < 1):
        print("h must be >= 1.")
        return _NA
    _lib.require_gpu()
    x = np.zeros(num)
    it = np.zeros(num, dtype=np.int32)
    h_a = _recycle(h, num)
    z_a = _recycle(z, num)
    _lib.lib().rpg_sp(_dp(x), _dp(h_a), _dp(z_a), _ci(num), _ip(it))
    if not track_iter:
        return x
    return {"samp": x, "iter": it}


def rpg(num=1, h=1, z=0.0):
    """LogitWrapper.R:104-121: PG(h, z) by the hybrid dispatch."""
    if np.any(np.asarray(h) <= 0):
        print("h must be > 0.")
        return _NA
    _lib.require_gpu()
    x = np.zeros(num)
    h_a = _recycle(h, num)
    z_a = _recycle(z, num)
    _lib.lib().rpg_hybrid(_dp(x), _dp(h_a), _dp(z_a), _ci(num))
    return x


def _check_parameters(y, n, m0, P0, R_X, C_X, samp, burn):
    """LogitWrapper.R:130-157."""
    P0 = np.asarray(P0)
    ok = [True] * 9
    ok[0] = bool(np.all(y >= 0))
    ok[1] = bool(np.all(n > 0))
    ok[2] = C_X == P0.shape[0]
    ok[3] = P0.ndim == 2 and C_X == P0.shape[1]
    ok[4] = (len(y) == len(n)) and (len(y) == R_X)
    ok[5] = C_X == np.size(m0)
    ok[6] = samp > 0
    ok[7] = burn >= 0
    ok[8] = bool(np.all(y <= 1))
    if not ok[0]:
        print("y must be >= 0.")
    if not ok[8]:
        print("y is a proportion; it must be <= 1.")
    if not ok[1]:
        print("n must be > 0.")
    if not ok[2]:
        print(f"col(X) != row(P0) {C_X} {P0.shape[0]}")
    if not ok[3]:
        print(f"col(X) != col(P0) {C_X} {P0.shape[-1]}")
    if not ok[4]:
        print(f"Dimensions do not conform for y, X, and n. len(y) = {len(y)} dim(x) = {R_X} {C_X} len(n) = {len(n)}")
    if not ok[5]:
        print(f"col(X) != length(m0) {C_X} {np.size(m0)}")
    if not ok[6]:
        print("samp must be > 0.")
    if not ok[7]:
        print("burn must be >=0.")
    return all(ok)


def _as_matrix(X):
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    return X


def logit_combine(y, X, n=None):
    """LogitWrapper.R:161-189: merge rows with identical covariates."""
    X = _as_matrix(X)
    N, P = X.shape
    y = np.asarray(y, dtype=np.float64).ravel()
    n = np.ones(len(y)) if n is None else np.asarray(n, dtype=np.float64).ravel()
    if not _check_parameters(y, n, np.zeros(P), np.zeros((P, P)), N, P, 1, 0):
        return -1
    _lib.require_gpu()
    tX = np.ascontiguousarray(X).copy()        # row-major N x P == column-major P x N = t(X)
    yb = y.copy()
    nb = n.copy()
    Nc = C.c_int(N)
    _lib.lib().combine(_dp(yb), _dp(tX), _dp(nb), C.byref(Nc), _ci(P))
    M = Nc.value
    return {"y": yb[:M].copy(), "X": tX.reshape(N, P)[:M].copy(), "n": nb[:M].copy()}


def logit(y, X, n=None, m0=None, P0=None, samp=1000, burn=500):
    """LogitWrapper.R:197-244.  Returns dict(w[samp x N'], beta[samp x P], y, X, n)."""
    X = _as_matrix(X)
    y = np.asarray(y, dtype=np.float64).ravel()
    n = np.ones(len(y)) if n is None else np.asarray(n, dtype=np.float64).ravel()
    new = logit_combine(y, X, n)
    if not isinstance(new, dict):
        return -1
    y, X, n = new["y"], new["X"], new["n"]
    N, P = X.shape
    m0 = np.zeros(P) if m0 is None else np.asarray(m0, dtype=np.float64).ravel()
    P0 = np.zeros((P, P)) if P0 is None else np.asarray(P0, dtype=np.float64)
    if not _check_parameters(y, n, m0, P0, N, P, samp, burn):
        return -1
    w = np.zeros((samp, N))          # memory == column-major N x samp
    beta = np.zeros((samp, P))       # memory == column-major P x samp
    tX = np.ascontiguousarray(X)
    P0f = np.asfortranarray(P0)
    Nc = C.c_int(N)
    _lib.lib().gibbs(_dp(w), _dp(beta), _dp(y), _dp(tX), _dp(n), _dp(m0),
                     P0f.ctypes.data_as(_lib.c_dp), C.byref(Nc), _ci(P), _ci(samp), _ci(burn))
    return {"w": w[:, :Nc.value], "beta": beta, "y": y, "X": X, "n": n}


def logit_EM(y, X, n=None, tol=1e-9, max_iter=100):
    """LogitWrapper.R:248-285.  Returns dict(beta, iter)."""
    X = _as_matrix(X)
    y = np.asarray(y, dtype=np.float64).ravel()
    n = np.ones(len(y)) if n is None else np.asarray(n, dtype=np.float64).ravel()
    new = logit_combine(y, X, n)
    if not isinstance(new, dict):
        return -1
    y, X, n = new["y"], new["X"], new["n"]
    N, P = X.shape
    if not _check_parameters(y, n, np.zeros(P), np.zeros((P, P)), N, P, 1, 0):
        return -1
    beta = np.zeros(P)
    it = C.c_int(int(max_iter))
    _lib.lib().EM(_dp(beta), _dp(y), _dp(np.ascontiguousarray(X)), _dp(n), _ci(N), _ci(P),
                  C.byref(C.c_double(tol)), C.byref(it))
    return {"beta": beta, "iter": it.value}


def _mult_check_parameters(y, X, n, m0, P0, samp, burn):
    """LogitWrapper.R:293-321."""
    ok = [True] * 8
    ok[0] = bool(np.all(y >= 0))
    ok[1] = bool(np.all(n > 0))
    ok[2] = y.shape[0] == len(n) and y.shape[0] == X.shape[0]
    ok[3] = samp > 0
    ok[4] = burn >= 0
    ok[5] = bool(np.all(y.sum(axis=1) <= 1))
    ok[6] = m0.ndim == 2 and y.shape[1] == m0.shape[1] and X.shape[1] == m0.shape[0]
    ok[7] = (P0.ndim == 3 and X.shape[1] == P0.shape[0] and X.shape[1] == P0.shape[1]
             and y.shape[1] == P0.shape[2])
    if not ok[0]:
        print("y must be >= 0.")
    if not ok[5]:
        print("y[i,] are proportions and must sum <= 1.")
    if not ok[1]:
        print("n must be > 0.")
    if not ok[2]:
        print(f"Dimensions do not conform for y, X, and n. dim(y) = {y.shape} dim(x) = {X.shape} len(n) = {len(n)}")
    if not ok[3]:
        print("samp must be > 0.")
    if not ok[4]:
        print("burn must be >=0.")
    if not ok[6]:
        print("m.0 does not conform.")
    if not ok[7]:
        print("P.0 does not conform.")
    return all(ok)


def mlogit_combine(y, X, n=None):
    """LogitWrapper.R:326-354."""
    X = _as_matrix(X)
    y = _as_matrix(y)
    N, P = X.shape
    U = y.shape[1]
    n = np.ones(y.shape[0]) if n is None else np.asarray(n, dtype=np.float64).ravel()
    if not _mult_check_parameters(y, X, n, np.zeros((P, U)), np.zeros((P, P, U)), 1, 0):
        return _NA
    _lib.require_gpu()
    ty = np.ascontiguousarray(y).copy()        # row-major N x U == column-major U x N = t(y)
    tX = np.ascontiguousarray(X).copy()
    nb = n.copy()
    Nc = C.c_int(N)
    _lib.lib().mult_combine(_dp(ty), _dp(tX), _dp(nb), C.byref(Nc), _ci(P), _ci(U + 1))
    M = Nc.value
    return {"y": ty.reshape(N, U)[:M].copy(), "X": tX.reshape(N, P)[:M].copy(), "n": nb[:M].copy()}


def mlogit(y, X, n=None, m_0=None, P_0=None, samp=1000, burn=500):
    """LogitWrapper.R:358-416.  Returns dict(w[samp x N' x (J-1)], beta[samp x P x (J-1)], y, X, n)."""
    X = _as_matrix(X)
    y = _as_matrix(y)
    n = np.ones(y.shape[0]) if n is None else np.asarray(n, dtype=np.float64).ravel()
    new = mlogit_combine(y, X, n)
    if not isinstance(new, dict):
        return _NA
    y, X, n = new["y"], new["X"], new["n"]
    N, P = X.shape
    U = y.shape[1]
    m_0 = np.zeros((P, U)) if m_0 is None else np.asarray(m_0, dtype=np.float64)
    P_0 = np.zeros((P, P, U)) if P_0 is None else np.asarray(P_0, dtype=np.float64)
    if not _mult_check_parameters(y, X, n, m_0, P_0, samp, burn):
        return _NA
    w = np.zeros((samp, U, N))        # memory == column-major N x U x samp
    beta = np.zeros((samp, U, P))     # memory == column-major P x U x samp
    Nc = C.c_int(N)
    m0f = np.asfortranarray(m_0)
    P0f = np.asfortranarray(P_0)
    _lib.lib().mult_gibbs(_dp(w), _dp(beta), _dp(np.ascontiguousarray(y)), _dp(np.ascontiguousarray(X)), _dp(n),
                          m0f.ctypes.data_as(_lib.c_dp), P0f.ctypes.data_as(_lib.c_dp),
                          C.byref(Nc), _ci(P), _ci(U + 1), _ci(samp), _ci(burn))
    return {"w": w.transpose(0, 2, 1), "beta": beta.transpose(0, 2, 1), "y": y, "X": X, "n": n}
