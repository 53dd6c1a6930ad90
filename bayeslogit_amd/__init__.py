"""bayeslogit_amd -- MI355X-native Polya-Gamma sampler and logistic Gibbs sweep.

Host-side mirror of the reference's R API (Code/R/LogitWrapper.R): same function
names (with `.` -> `_`), defaults, argument meaning, return shapes and error
behaviour, calling the same `.C`-style entry points of libbayeslogit_hip.so
(include/bayeslogit_hip.h) that the R file would.  All compute runs in HIP
kernels; there is no CPU implementation in this package.
"""
from ._lib import BayesLogitError, lib, require_gpu  # noqa: F401
from .api import (  # noqa: F401
    logit,
    logit_combine,
    logit_EM,
    mlogit,
    mlogit_combine,
    rpg,
    rpg_alt,
    rpg_devroye,
    rpg_gamma,
    rpg_sp,
    set_seed,
)

__all__ = [
    "rpg", "rpg_devroye", "rpg_alt", "rpg_sp", "rpg_gamma", "logit", "logit_EM", "logit_combine",
    "mlogit", "mlogit_combine", "set_seed", "BayesLogitError",
]
