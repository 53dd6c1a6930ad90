"""ctypes loader for the CPU oracle (oracle/liboracle.so).

Test infrastructure only: the oracle is the CHECKER.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg load it; the product
package bayeslogit_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

c_d = C.c_double
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_u64 = C.c_uint64
c_u32 = C.c_uint32
c_i64 = C.c_int64


class BlRng(C.Structure):
    _fields_ = [("key", c_u32 * 2), ("ctr", c_u32 * 4), ("buf", c_u32 * 4),
                ("pos", C.c_int), ("nunif", c_u64)]


def build():
    """Compile liboracle.so.  Called by tests/conftest.py at configure time (before any test has touched the GPU)
    and by __graft_entry__.build(); never from lib(): a process that has initialised the GPU (bench.py, smoke())
    must not start other programs on this pool."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it first with `make -C oracle liboracle.so` "
                           "(__graft_entry__.build() does); lib() does not compile anything")
    L = C.CDLL(path)
    rp = C.POINTER(BlRng)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("bl_philox4x32_10", None, C.POINTER(c_u32), C.POINTER(c_u32), C.POINTER(c_u32))
    sig("bl_chain_key", c_u64, c_u64, c_u32)
    sig("bl_rng_init", None, rp, c_u64, c_u64, c_u32, c_u32)
    sig("bl_unif", c_d, rp)
    sig("bl_expon_rate", c_d, rp, c_d)
    sig("bl_norm", c_d, rp, c_d, c_d)
    sig("bl_gamma_scale", c_d, rp, c_d, c_d)
    sig("bl_igauss", c_d, rp, c_d, c_d)
    sig("bl_ltgamma", c_d, rp, c_d, c_d, c_d)
    sig("bl_rtinvchi2", c_d, rp, c_d, c_d)
    sig("bl_tnorm", c_d, rp, c_d, c_d)
    sig("bl_flat", c_d, rp, c_d, c_d)
    sig("bl_p_norm", c_d, c_d, C.c_int)
    sig("bl_p_gamma_rate", c_d, c_d, c_d, c_d)
    sig("bl_p_igauss", c_d, c_d, c_d, c_d)
    sig("bl_pg_draw_devroye", c_d, C.c_int, c_d, rp)
    sig("bl_pg_a", c_d, C.c_int, c_d)
    sig("bl_pg_mass_texpon", c_d, c_d)
    sig("bl_pg_m1", c_d, c_d, c_d)
    sig("bl_pg_m2", c_d, c_d, c_d)
    sig("bl_alt_a_coef", c_d, C.c_int, c_d, c_d)
    sig("bl_alt_g_tilde", c_d, c_d, c_d, c_d)
    sig("bl_alt_w_left", c_d, c_d, c_d, c_d)
    sig("bl_alt_w_right", c_d, c_d, c_d, c_d)
    sig("bl_y_eval", c_d, c_d)
    sig("bl_ydy_eval", None, c_d, c_dp, c_dp)
    sig("bl_fdf_eval", None, c_d, c_d, c_dp, c_dp)
    sig("bl_v_eval", c_d, c_d)
    sig("bl_sp_y_func", c_d, c_d)
    sig("bl_sp_approx", c_d, c_d, c_d, c_d)
    sig("bl_sp_tangent_to_eta", None, c_d, c_d, c_d, c_dp, c_dp)
    sig("bl_o_rpg_devroye", None, c_dp, c_ip, c_dp, c_i64, c_u64, c_u32, c_u64)
    sig("bl_o_rpg_devroye_omp", None, c_dp, c_ip, c_dp, c_i64, c_u64, c_u32, c_u64, C.c_int, C.c_int)
    sig("bl_o_rpg_alt", None, c_dp, c_dp, c_dp, c_i64, c_u64, c_u32, c_u64)
    sig("bl_o_rpg_sp", None, c_dp, c_dp, c_dp, c_i64, c_ip, c_u64, c_u32, c_u64)
    sig("bl_o_rpg_gamma", None, c_dp, c_dp, c_dp, c_i64, C.c_int, c_u64, c_u32, c_u64)
    sig("bl_o_rpg_hybrid", None, c_dp, c_dp, c_dp, c_i64, c_u64, c_u32, c_u64)
    c_u32p = C.POINTER(c_u32)
    sig("bl_o_rpg_alt_attempt", None, c_dp, c_dp, c_dp, c_i64, c_u64, c_u32, c_u64, c_u32p)
    sig("bl_o_rpg_sp_attempt", None, c_dp, c_dp, c_dp, c_i64, c_ip, c_u64, c_u32, c_u64, c_u32p)
    sig("bl_o_rpg_hybrid_attempt", None, c_dp, c_dp, c_dp, c_i64, c_u64, c_u32, c_u64)
    sig("bl_pg_devroye_literal_census", None, c_d, c_i64, c_u64, C.POINTER(c_i64), C.POINTER(c_i64))
    sig("bl_sp_vlk", None, c_d, c_d, c_dp, c_dp, c_dp)
    sig("bl_upper_gamma_cf", c_d, c_d, c_d)
    sig("bl_o_rpg_hybrid_omp", None, c_dp, c_dp, c_dp, c_i64, c_u64, c_u32, c_u64, C.c_int)
    sig("bl_o_max_threads", C.c_int)
    sig("bl_o_gibbs", C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_i64, C.c_int,
        C.c_int, C.c_int, c_u64, C.c_int, c_u64)
    sig("bl_o_sweep_partial", None, c_dp, c_dp, c_dp, c_dp, c_dp, c_i64, C.c_int, c_u64, c_u32, c_u64)
    sig("bl_o_draw_beta", None, c_dp, c_dp, c_dp, c_dp, C.c_int, c_u64, c_u32, C.c_int)
    sig("bl_o_set_bP", None, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_i64, C.c_int)
    sig("bl_o_EM", C.c_int, c_dp, c_dp, c_dp, c_dp, c_i64, C.c_int, c_d, C.c_int)
    sig("bl_o_combine", c_i64, c_dp, c_dp, c_dp, c_i64, C.c_int)
    sig("bl_o_mult_combine", c_i64, c_dp, c_dp, c_dp, c_i64, C.c_int, C.c_int)
    sig("bl_o_mult_gibbs", C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_i64, C.c_int,
        C.c_int, C.c_int, C.c_int, c_u64)
    sig("bl_chol_upper", C.c_int, c_dp, c_dp, C.c_int)
    sig("bl_chol_lower", C.c_int, c_dp, c_dp, C.c_int)
    _LIB = L
    return L


def dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


def _f64(a, n=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if n is not None and a.size != n:
        a = np.ascontiguousarray(np.resize(a, n))
    return a


def rng(seed, idx=0, domain=0, epoch=0):
    r = BlRng()
    lib().bl_rng_init(C.byref(r), seed, idx, domain, epoch)
    return r


def chain_key(seed, call=0):
    """Key of the chain the call-th gibbs()/mult_gibbs() of the .C boundary starts after set_seed(seed)."""
    return lib().bl_chain_key(seed, call)


def philox(ctr, key):
    c = (c_u32 * 4)(*ctr)
    k = (c_u32 * 2)(*key)
    o = (c_u32 * 4)()
    lib().bl_philox4x32_10(c, k, o)
    return list(o)


def rpg_devroye(num, n, z, seed, epoch=0, idx0=0, threads=0, literal=False):
    """literal=False: the attempt form (what the HIP path computes, draw for draw);
    literal=True: the reference's loops call for call (same distribution, other stream use)."""
    n = np.ascontiguousarray(np.resize(np.asarray(n, dtype=np.int32), num))
    z = _f64(z, num)
    x = np.zeros(num)
    if threads or literal:
        lib().bl_o_rpg_devroye_omp(dp(x), ip(n), dp(z), num, seed, epoch, idx0, max(1, threads), int(literal))
    else:
        lib().bl_o_rpg_devroye(dp(x), ip(n), dp(z), num, seed, epoch, idx0)
    return x


def _rpg_h(fn, num, h, z, seed, epoch, idx0, *extra):
    h = _f64(h, num)
    z = _f64(z, num)
    x = np.zeros(num)
    fn(dp(x), dp(h), dp(z), num, *extra, seed, epoch, idx0)
    return x


def rpg_alt(num, h, z, seed, epoch=0, idx0=0, literal=False, blocks=False):
    """literal=False: the attempt form (what the HIP path computes, draw for draw);
    literal=True: PolyaGammaAlt.cpp's loops call for call (same distribution, other stream use)."""
    if literal:
        return _rpg_h(lib().bl_o_rpg_alt, num, h, z, seed, epoch, idx0)
    h = _f64(h, num)
    z = _f64(z, num)
    x = np.zeros(num)
    nb = np.zeros(num, dtype=np.uint32)
    lib().bl_o_rpg_alt_attempt(dp(x), dp(h), dp(z), num, seed, epoch, idx0, nb.ctypes.data_as(C.POINTER(c_u32)))
    return (x, nb) if blocks else x


def rpg_hybrid(num, h, z, seed, epoch=0, idx0=0, threads=0, literal=False):
    """literal=False: every branch in the form the HIP path computes; literal=True: Alt and SP by the
    reference's loops (Devroye stays in its attempt form; rpg_devroye(literal=True) has the loops)."""
    if not literal:
        return _rpg_h(lib().bl_o_rpg_hybrid_attempt, num, h, z, seed, epoch, idx0)   # OpenMP inside: same draws
    if threads:
        h = _f64(h, num)
        z = _f64(z, num)
        x = np.zeros(num)
        lib().bl_o_rpg_hybrid_omp(dp(x), dp(h), dp(z), num, seed, epoch, idx0, threads)
        return x
    return _rpg_h(lib().bl_o_rpg_hybrid, num, h, z, seed, epoch, idx0)


def rpg_gamma(num, h, z, seed, trunc=200, epoch=0, idx0=0):
    return _rpg_h(lib().bl_o_rpg_gamma, num, h, z, seed, epoch, idx0, trunc)


def rpg_sp(num, h, z, seed, epoch=0, idx0=0, literal=False, blocks=False):
    h = _f64(h, num)
    z = _f64(z, num)
    x = np.zeros(num)
    it = np.zeros(num, dtype=np.int32)
    if literal:
        lib().bl_o_rpg_sp(dp(x), dp(h), dp(z), num, ip(it), seed, epoch, idx0)
        return x, it
    nb = np.zeros(num, dtype=np.uint32)
    lib().bl_o_rpg_sp_attempt(dp(x), dp(h), dp(z), num, ip(it), seed, epoch, idx0, nb.ctypes.data_as(C.POINTER(c_u32)))
    return (x, it, nb) if blocks else (x, it)


def devroye_census(z, ndraws, seed):
    counts = (c_i64 * 4)()
    nprop = c_i64(0)
    lib().bl_pg_devroye_literal_census(z, ndraws, seed, counts, C.byref(nprop))
    return list(counts), nprop.value


def sp_vlk(x):
    v, L, k = c_d(), c_d(), c_d()
    lib().bl_sp_vlk(x, float(np.log(x)), C.byref(v), C.byref(L), C.byref(k))
    return v.value, L.value, k.value


def gibbs(y, X, n, m0, P0, samp, burn, seed, constrain=1, store_w=True, idx0=0):
    """X is N x P (row-major numpy) == tX P x N column-major. Returns (w samp x N, beta samp x P)."""
    X = _f64(X)
    N, P = X.shape
    y = _f64(y)
    n = _f64(n)
    m0 = _f64(m0)
    P0 = np.asfortranarray(np.asarray(P0, dtype=np.float64))
    w = np.zeros((samp, N)) if store_w else None
    beta = np.zeros((samp, P))
    rc = lib().bl_o_gibbs(dp(w), dp(beta), dp(y), dp(X), dp(n), dp(m0),
                          P0.ctypes.data_as(c_dp), N, P, samp, burn, seed, constrain, idx0)
    assert rc == 0
    return w, beta


def sweep_partial(X, n, beta, seed, sweep, idx0=0):
    X = _f64(X)
    N, P = X.shape
    n = _f64(n)
    beta = _f64(beta)
    PP = np.zeros((P, P))
    w = np.zeros(N)
    lib().bl_o_sweep_partial(dp(PP), dp(w), dp(X), dp(n), dp(beta), N, P, seed, sweep, idx0)
    return PP, w


def draw_beta(PP, bP, beta_prev, seed, sweep, constrain):
    PP = np.asfortranarray(np.asarray(PP, dtype=np.float64))
    P = PP.shape[0]
    bP = _f64(bP)
    beta_prev = _f64(beta_prev)
    out = np.zeros(P)
    lib().bl_o_draw_beta(dp(out), PP.ctypes.data_as(c_dp), dp(bP), dp(beta_prev), P, seed, sweep, constrain)
    return out


def set_bP(y, X, n, m0, P0):
    X = _f64(X)
    N, P = X.shape
    P0 = np.asfortranarray(np.asarray(P0, dtype=np.float64))
    out = np.zeros(P)
    lib().bl_o_set_bP(dp(out), dp(_f64(y)), dp(X), dp(_f64(n)), dp(_f64(m0)), P0.ctypes.data_as(c_dp), N, P)
    return out


def em(y, X, n, tol=1e-9, max_iter=100):
    X = _f64(X)
    N, P = X.shape
    beta = np.zeros(P)
    it = lib().bl_o_EM(dp(beta), dp(_f64(y)), dp(X), dp(_f64(n)), N, P, tol, max_iter)
    return beta, it


def combine(y, X, n):
    X = _f64(X).copy()
    N, P = X.shape
    y = _f64(y).copy()
    n = _f64(n).copy()
    M = lib().bl_o_combine(dp(y), dp(X), dp(n), N, P)
    return y[:M].copy(), X[:M].copy(), n[:M].copy()


def mult_combine(y, X, n):
    """y is N x (J-1)."""
    X = _f64(X).copy()
    y = _f64(y).copy()
    N, P = X.shape
    J = y.shape[1] + 1
    n = _f64(n).copy()
    M = lib().bl_o_mult_combine(dp(y), dp(X), dp(n), N, P, J)
    return y[:M].copy(), X[:M].copy(), n[:M].copy()


def mult_gibbs(y, X, n, m0, P0, samp, burn, seed, store_w=True):
    """y N x (J-1); m0 P x (J-1); P0 P x P x (J-1). Returns w (samp,N,J-1), beta (samp,P,J-1)."""
    X = _f64(X)
    y = _f64(y)
    N, P = X.shape
    U = y.shape[1]
    m0f = np.asfortranarray(np.asarray(m0, dtype=np.float64).reshape(P, U))
    P0f = np.asfortranarray(np.asarray(P0, dtype=np.float64).reshape(P, P, U))
    w = np.zeros((samp, U, N)) if store_w else None   # memory order of N x U x samp column-major
    beta = np.zeros((samp, U, P))
    rc = lib().bl_o_mult_gibbs(dp(w), dp(beta), dp(y), dp(X), dp(_f64(n)),
                               m0f.ctypes.data_as(c_dp), P0f.ctypes.data_as(c_dp),
                               N, P, U + 1, samp, burn, seed)
    assert rc == 0
    return (None if w is None else w.transpose(0, 2, 1)), beta.transpose(0, 2, 1)
