"""ctypes binding of libbayeslogit_hip.so (the C ABI of include/bayeslogit_hip.h and of the diagnostic header include/bayeslogit_hip_diag.h).

There is no CPU implementation behind this package: if the HIP library is not
built, or no GPU is present when a compute entry point is called, the call
fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BAYESLOGIT_LIB", os.path.join(_HERE, "libbayeslogit_hip.so"))   # override: A/B builds
_LIB = None

c_d = C.c_double
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_u64 = C.c_uint64
c_u32 = C.c_uint32
c_i64 = C.c_int64
c_vp = C.c_void_p

# every symbol include/bayeslogit_hip.h and include/bayeslogit_hip_diag.h declare: name -> (restype, argtypes)
SIGNATURES = {
    "bl_last_error": (C.c_char_p, []),
    "bl_last_sampler_flags": (C.c_int, []),
    "bl_device_count": (C.c_int, []),
    "bl_set_device": (C.c_int, [C.c_int]),
    "bl_set_seed": (None, [c_u64]),
    "bl_get_seed": (c_u64, []),
    "bl_set_seed_from_unif": (None, [c_dp]),
    "bl_get_epoch": (c_u32, []),
    "bl_set_constrain": (None, [C.c_int]),
    "bl_set_constrain_R": (None, [c_ip]),
    "bl_set_device_R": (None, [c_ip, c_ip]),
    "bl_set_sweep_mode": (None, [C.c_int]),
    "bl_diag_sweep_deferred": (C.c_int, [C.POINTER(c_u64)]),
    "bl_diag_beta_sweeps": (None, [C.c_int]),
    "rpg_gamma": (None, [c_dp, c_dp, c_dp, c_ip, c_ip]),
    "rpg_devroye": (None, [c_dp, c_ip, c_dp, c_ip]),
    "rpg_alt": (None, [c_dp, c_dp, c_dp, c_ip]),
    "rpg_sp": (None, [c_dp, c_dp, c_dp, c_ip, c_ip]),
    "rpg_hybrid": (None, [c_dp, c_dp, c_dp, c_ip]),
    "gibbs": (None, [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip]),
    "EM": (None, [c_dp, c_dp, c_dp, c_dp, c_ip, c_ip, c_dp, c_ip]),
    "combine": (None, [c_dp, c_dp, c_dp, c_ip, c_ip]),
    "mult_gibbs": (None, [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip, c_ip]),
    "mult_combine": (None, [c_dp, c_dp, c_dp, c_ip, c_ip, c_ip]),
    "bl_sync_status": (C.c_int, [c_vp]),
    "bl_rpg_devroye_dev": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_i64, c_u64, c_u32, c_u64, c_vp]),
    "bl_rpg_hybrid_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_u64, c_u32, c_u64, c_vp]),
    "bl_rpg_alt_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_u64, c_u32, c_u64, c_vp]),
    "bl_rpg_sp_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_u64, c_u32, c_u64, c_vp]),
    "bl_rpg_gamma_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, c_u64, c_u32, c_u64, c_vp]),
    "bl_diag_mfma_f64_dev": (C.c_int, [c_vp, c_i64, C.c_int, C.c_int, C.POINTER(C.c_double), c_vp]),
    "bl_diag_mfma_f64_small_dev": (C.c_int, [c_vp, c_i64, C.c_int, C.c_int, C.POINTER(C.c_double), c_vp]),
    "bl_diag_count_blocks_dev": (C.c_int, [c_vp, c_vp, c_i64, c_u64, c_u32, c_u64, c_vp, c_vp]),
    "bl_diag_rpg_hybrid_class_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, c_u64, c_u32, c_u64, c_vp]),
    "bl_diag_sp_vlk_dev": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "bl_fill_unif_dev": (C.c_int, [c_vp, c_i64, c_d, c_d, c_u64, c_u32, c_u64, c_vp]),
    "bl_fill_norm_dev": (C.c_int, [c_vp, c_i64, c_d, c_d, c_u64, c_u32, c_u64, c_vp]),
    "bl_fill_shape_dev": (C.c_int, [c_vp, c_i64, C.c_int, c_u64, c_u32, c_u64, c_vp]),
    "bl_fill_logit_y_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, c_u64, c_u32, c_u64, c_vp]),
    "bl_gibbs_create": (C.c_int, [C.POINTER(c_vp), c_i64, C.c_int, c_u64, c_u64, c_vp]),
    "bl_gibbs_destroy": (None, [c_vp]),
    "bl_gibbs_set_data": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "bl_gibbs_set_prior": (C.c_int, [c_vp, c_dp, c_dp]),
    "bl_gibbs_set_beta": (C.c_int, [c_vp, c_dp]),
    "bl_gibbs_set_bp_local": (C.c_int, [c_vp]),
    "bl_gibbs_finish_bp": (C.c_int, [c_vp]),
    "bl_gibbs_chain_start": (C.c_int, [c_vp]),
    "bl_gibbs_sweep_local": (C.c_int, [c_vp, c_u32, c_vp]),
    "bl_gibbs_draw_beta": (C.c_int, [c_vp, c_u32, C.c_int]),
    "bl_gibbs_em_local": (C.c_int, [c_vp]),
    "bl_gibbs_em_solve": (C.c_int, [c_vp, c_dp]),
    "bl_gibbs_pp_ptr": (c_vp, [c_vp]),
    "bl_gibbs_bp_ptr": (c_vp, [c_vp]),
    "bl_gibbs_beta_ptr": (c_vp, [c_vp]),
    "bl_gibbs_get_beta": (C.c_int, [c_vp, c_dp]),
    "bl_gibbs_run": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, c_dp, c_vp]),
    "bl_gibbs_run_stream": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp]),
}


class GibbsStats(C.Structure):
    """bl_gibbs_stats of include/bayeslogit_hip.h"""
    _fields_ = [("beta_mean_host", c_dp), ("beta_var_host", c_dp), ("w_mean_dev", c_vp), ("w_var_dev", c_vp)]


W_NONE, W_LAST, W_ALL = 0, 1, 2


class BayesLogitError(RuntimeError):
    pass


def lib():
    """Load the HIP library; raise if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise BayesLogitError(
            f"{LIB_PATH} is missing: build it with `python -m bayeslogit_amd.build` "
            "(bayeslogit_amd has no CPU implementation)")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)       # AttributeError here = header/library mismatch
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().bl_last_error()
        raise BayesLogitError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def require_gpu():
    if lib().bl_device_count() < 1:
        raise BayesLogitError("no HIP device visible; bayeslogit_amd has no CPU implementation")
