"""Device-resident entry points (Part 2 of include/bayeslogit_hip.h) on torch tensors.

torch is used only as the owner of device memory and of the HIP stream; the
kernels are the library's.  Every function launches on torch's current stream.
"""
import ctypes as C

import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f64(t, name):
    if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float64 CUDA tensor")
    return t


def sync_status():
    _lib.check(_lib.lib().bl_sync_status(_stream()), "bl_sync_status")


def rpg_devroye(z, n=1, seed=0, epoch=0, idx0=0, out=None):
    """x_i ~ PG(n_i, z_i), n integer (tensor int32 or scalar)."""
    _f64(z, "z")
    x = torch.empty_like(z) if out is None else _f64(out, "out")
    nvec = None
    nscalar = 1
    if torch.is_tensor(n):
        if n.dtype != torch.int32 or not n.is_cuda:
            raise ValueError("n must be an int32 CUDA tensor or an int")
        nvec = n
    else:
        nscalar = int(n)
    _lib.check(_lib.lib().bl_rpg_devroye_dev(_ptr(x), _ptr(nvec), nscalar, _ptr(z), z.numel(), seed, epoch, idx0,
                                             _stream()), "bl_rpg_devroye_dev")
    return x


def _rpg_h(fn, name, h, z, seed, epoch, idx0, out, *extra):
    _f64(z, "z")
    _f64(h, "h")
    x = torch.empty_like(z) if out is None else _f64(out, "out")
    _lib.check(fn(_ptr(x), _ptr(h), _ptr(z), z.numel(), *extra, seed, epoch, idx0, _stream()), name)
    return x


def rpg_hybrid(h, z, seed=0, epoch=0, idx0=0, out=None):
    return _rpg_h(_lib.lib().bl_rpg_hybrid_dev, "bl_rpg_hybrid_dev", h, z, seed, epoch, idx0, out)


def rpg_alt(h, z, seed=0, epoch=0, idx0=0, out=None):
    return _rpg_h(_lib.lib().bl_rpg_alt_dev, "bl_rpg_alt_dev", h, z, seed, epoch, idx0, out)


def rpg_gamma(h, z, trunc=200, seed=0, epoch=0, idx0=0, out=None):
    return _rpg_h(_lib.lib().bl_rpg_gamma_dev, "bl_rpg_gamma_dev", h, z, seed, epoch, idx0, out, int(trunc))


def rpg_sp(h, z, seed=0, epoch=0, idx0=0, out=None, iters=None):
    return _rpg_h(_lib.lib().bl_rpg_sp_dev, "bl_rpg_sp_dev", h, z, seed, epoch, idx0, out, _ptr(iters))


def mfma_f64_sustained_tflops(waves_per_simd=2, iters=4000, small=False):
    """Measured rate of a register-only loop of v_mfma_f64_16x16x4_f64 (bl_diag_mfma_f64_dev) or, small=True, of
    v_mfma_f64_4x4x4_4b_f64 (bl_diag_mfma_f64_small_dev), TFLOP/s."""
    work = torch.empty(1024 * 8 * 256, dtype=torch.float64, device="cuda")
    fl = C.c_double(0.0)
    best = 0.0
    fn = _lib.lib().bl_diag_mfma_f64_small_dev if small else _lib.lib().bl_diag_mfma_f64_dev
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(fn(_ptr(work), work.numel(), waves_per_simd, iters, C.byref(fl), _stream()), "bl_diag_mfma_f64_dev")
        e1.record()
        e1.synchronize()
        if rep:
            best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


CLASS_NAMES = ("zero", "sum_of_gammas", "devroye", "alternating_series", "saddle_point", "normal_approximation",
               "devroye_wide_z")          # the last: the part of "devroye" with |z|/2 >= 1/t (the other left-piece sampler)


def count_blocks(h, z, seed=0, epoch=0, idx0=0):
    """Per sampler class {observations, draws, Philox blocks (= proposal attempts)} of the draws rpg_hybrid(h, z) -- or, h None,
    rpg_devroye(z, n = 1) -- makes on these streams (bl_diag_count_blocks_dev: an exact replay, one observation per lane)."""
    out = torch.zeros(21, dtype=torch.int64, device=z.device)
    _lib.check(_lib.lib().bl_diag_count_blocks_dev(_ptr(h), _ptr(_f64(z, "z")), z.numel(), seed, epoch, idx0, _ptr(out), _stream()),
               "bl_diag_count_blocks_dev")
    o = out.cpu().numpy().reshape(7, 3)
    return {CLASS_NAMES[c]: {"observations": int(o[c, 0]), "draws": int(o[c, 1]), "blocks": int(o[c, 2])}
            for c in range(7) if o[c, 0] > 0}


def rpg_hybrid_class(h, z, cls, seed=0, epoch=0, idx0=0, out=None):
    """ONE class pass of rpg_hybrid alone (bl_diag_rpg_hybrid_class_dev): only that class's elements of `out` are written."""
    x = torch.empty_like(z) if out is None else _f64(out, "out")
    _lib.check(_lib.lib().bl_diag_rpg_hybrid_class_dev(_ptr(x), _ptr(_f64(h, "h")), _ptr(_f64(z, "z")), z.numel(), int(cls), seed,
                                                       epoch, idx0, _stream()), "bl_diag_rpg_hybrid_class_dev")
    return x


def fill_unif(out, lo, hi, seed, epoch=0, idx0=0):
    _lib.check(_lib.lib().bl_fill_unif_dev(_ptr(_f64(out, "out")), out.numel(), lo, hi, seed, epoch, idx0, _stream()),
               "bl_fill_unif_dev")
    return out


def fill_norm(out, mean, sd, seed, epoch=0, idx0=0):
    _lib.check(_lib.lib().bl_fill_norm_dev(_ptr(_f64(out, "out")), out.numel(), mean, sd, seed, epoch, idx0,
                                           _stream()), "bl_fill_norm_dev")
    return out


def fill_shape(out, kmax, seed, epoch=0, idx0=0):
    _lib.check(_lib.lib().bl_fill_shape_dev(_ptr(_f64(out, "out")), out.numel(), int(kmax), seed, epoch, idx0,
                                            _stream()), "bl_fill_shape_dev")
    return out


def fill_logit_y(y, X, beta, seed, epoch=0, idx0=0):
    """X: (N, P) row-major tensor == tX P x N column-major."""
    N, P = X.shape
    _lib.check(_lib.lib().bl_fill_logit_y_dev(_ptr(_f64(y, "y")), _ptr(_f64(X, "X")), _ptr(_f64(beta, "beta")), N, P,
                                              seed, epoch, idx0, _stream()), "bl_fill_logit_y_dev")
    return y


def set_sweep_mode(single_pass=True):
    """P = 64 Gibbs sweeps read X once (default) or in two streaming passes (bl_set_sweep_mode)."""
    _lib.lib().bl_set_sweep_mode(1 if single_pass else 0)


def set_beta_sweeps(row_split=True):
    """Which of the two (bit-identical) kernels runs the constrained sweeps of 64 < P <= 256 (bl_diag_beta_sweeps)."""
    _lib.lib().bl_diag_beta_sweeps(1 if row_split else 0)


def sweep_deferred_rows():
    """Rows the single-pass sweep handed to the full sampler since the last call (synchronises the device)."""
    v = C.c_uint64(0)
    _lib.check(_lib.lib().bl_diag_sweep_deferred(C.byref(v)), "bl_diag_sweep_deferred")
    return int(v.value)


class GibbsShard:
    """This rank's rows of the logistic Gibbs problem (bl_gibbs handle).

    X: (N_local, P) row-major float64 CUDA tensor (== tX, P x N_local column-major),
    y, n: (N_local,) float64 CUDA tensors.  idx0 = global index of local row 0.
    """

    def __init__(self, X, y, n, seed, idx0=0):
        self.X, self.y, self.n = _f64(X, "X"), y, _f64(n, "n")
        if y is not None:
            _f64(y, "y")
        self.N, self.P = X.shape
        self.h = C.c_void_p()
        L = _lib.lib()
        _lib.check(L.bl_gibbs_create(C.byref(self.h), self.N, self.P, idx0, seed, _stream()), "bl_gibbs_create")
        _lib.check(L.bl_gibbs_set_data(self.h, _ptr(X), _ptr(y), _ptr(n)), "bl_gibbs_set_data")

    def close(self):
        if self.h:
            _lib.lib().bl_gibbs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, ptr, numel):
        # a float64 tensor aliasing library-owned device memory (for collectives)
        import numpy as np  # noqa: F401
        return _DevView.as_tensor(ptr, numel)

    def set_prior(self, m0, P0):
        import numpy as np
        m0 = np.ascontiguousarray(m0, dtype=np.float64)
        P0 = np.asfortranarray(P0, dtype=np.float64)
        _lib.check(_lib.lib().bl_gibbs_set_prior(self.h, m0.ctypes.data_as(_lib.c_dp), P0.ctypes.data_as(_lib.c_dp)),
                   "bl_gibbs_set_prior")

    def set_beta(self, beta):
        import numpy as np
        b = np.ascontiguousarray(beta, dtype=np.float64)
        _lib.check(_lib.lib().bl_gibbs_set_beta(self.h, b.ctypes.data_as(_lib.c_dp)), "bl_gibbs_set_beta")

    def get_beta(self):
        import numpy as np
        b = np.zeros(self.P)
        _lib.check(_lib.lib().bl_gibbs_get_beta(self.h, b.ctypes.data_as(_lib.c_dp)), "bl_gibbs_get_beta")
        return b

    def pp(self):
        return _DevView.as_tensor(_lib.lib().bl_gibbs_pp_ptr(self.h), self.P * self.P)

    def bp(self):
        return _DevView.as_tensor(_lib.lib().bl_gibbs_bp_ptr(self.h), self.P)

    def beta(self):
        return _DevView.as_tensor(_lib.lib().bl_gibbs_beta_ptr(self.h), self.P)

    def set_bp_local(self):
        _lib.check(_lib.lib().bl_gibbs_set_bp_local(self.h), "bl_gibbs_set_bp_local")

    def finish_bp(self):
        _lib.check(_lib.lib().bl_gibbs_finish_bp(self.h), "bl_gibbs_finish_bp")

    def chain_start(self):
        """A chain driven sweep by sweep begins here (bl_gibbs_chain_start)."""
        _lib.check(_lib.lib().bl_gibbs_chain_start(self.h), "bl_gibbs_chain_start")

    def sweep_local(self, sweep, w_out=None):
        _lib.check(_lib.lib().bl_gibbs_sweep_local(self.h, sweep, _ptr(w_out)), "bl_gibbs_sweep_local")

    def draw_beta(self, sweep, constrain=1):
        _lib.check(_lib.lib().bl_gibbs_draw_beta(self.h, sweep, int(constrain)), "bl_gibbs_draw_beta")

    def em_local(self):
        _lib.check(_lib.lib().bl_gibbs_em_local(self.h), "bl_gibbs_em_local")

    def em_solve(self):
        d = C.c_double(0.0)
        _lib.check(_lib.lib().bl_gibbs_em_solve(self.h, C.byref(d)), "bl_gibbs_em_solve")
        return d.value

    def run(self, samp, burn, constrain=1, w_out=None):
        """Whole single-GPU chain; returns beta (samp, P) numpy."""
        import numpy as np
        beta = np.zeros((samp, self.P))
        _lib.check(_lib.lib().bl_gibbs_run(self.h, samp, burn, int(constrain), beta.ctypes.data_as(_lib.c_dp),
                                           _ptr(w_out)), "bl_gibbs_run")
        return beta

    def run_stream(self, samp, burn, constrain=1, thin=1, store_w="none", moments=False):
        """The chain of run() with streamed / reduced outputs (bl_gibbs_run_stream).
        store_w: "none" | "last" (device tensor, N) | "all" (host numpy, samp x N, streamed through a ring).
        moments: also return running mean / variance of beta (numpy) and omega (device tensors) over the
        samp sweeps, accumulated on the device.  Returns a dict."""
        import ctypes as C
        import numpy as np
        nkeep = (samp + thin - 1) // thin
        beta = np.zeros((nkeep, self.P))
        out = {"beta": beta}
        mode = {"none": _lib.W_NONE, "last": _lib.W_LAST, "all": _lib.W_ALL}[store_w]
        wptr = None
        if store_w == "last":
            out["w"] = torch.empty(self.N, dtype=torch.float64, device="cuda")
            wptr = _ptr(out["w"])
        elif store_w == "all":
            out["w"] = np.zeros((samp, self.N))
            wptr = out["w"].ctypes.data_as(C.c_void_p)
        st = None
        if moments:
            out["beta_mean"], out["beta_var"] = np.zeros(self.P), np.zeros(self.P)
            out["w_mean"] = torch.empty(self.N, dtype=torch.float64, device="cuda")
            out["w_var"] = torch.empty(self.N, dtype=torch.float64, device="cuda")
            st = _lib.GibbsStats(out["beta_mean"].ctypes.data_as(_lib.c_dp), out["beta_var"].ctypes.data_as(_lib.c_dp),
                                 _ptr(out["w_mean"]), _ptr(out["w_var"]))
        _lib.check(_lib.lib().bl_gibbs_run_stream(self.h, samp, burn, int(constrain), int(thin),
                                                  beta.ctypes.data_as(_lib.c_dp), mode, wptr,
                                                  C.byref(st) if st is not None else None), "bl_gibbs_run_stream")
        return out


class _DevView:
    """float64 torch tensor over a raw device pointer (no copy, no ownership)."""

    def __init__(self, ptr, numel):
        self.__cuda_array_interface__ = {
            "shape": (numel,), "typestr": "<f8", "data": (int(ptr), False), "version": 2, "strides": None,
        }

    @staticmethod
    def as_tensor(ptr, numel):
        return torch.as_tensor(_DevView(ptr, numel), device="cuda")
