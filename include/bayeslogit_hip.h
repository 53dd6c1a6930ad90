/* bayeslogit_hip.h -- C ABI of libbayeslogit_hip.so (MI355X / gfx950).
 *
 * Part 1 is the drop-in boundary: the exact `extern "C"` table that BayesLogit's
 * R layer reaches through .C("name", ..., PACKAGE="BayesLogit")
 * (reference Code/C/LogitWrapper.h:23-64, called from Code/R/LogitWrapper.R:29,49,69,
 * 92,118,177,229,277,342,395).  Same names, same pointer signatures, same layouts
 * (column-major; tX is P x N), same error behaviour (no return code: a message is
 * printed and the call returns, LogitWrapper.cpp:226-229).  Buffers are caller-owned
 * HOST memory, as .C hands them over.
 *
 * Part 2 is the device-resident form of the same operations: plain pointers and
 * sizes, pointers are DEVICE pointers, every call takes the counter-RNG
 * coordinates (seed, epoch, idx0) explicitly and returns a status.  This is what a
 * multi-GPU driver (one process per GPU) and the benchmark call.
 *
 * No torch types appear anywhere in this header.
 */
#ifndef BAYESLOGIT_HIP_H
#define BAYESLOGIT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
enum {
  BL_OK = 0,
  BL_ERR_NO_DEVICE = 1,   /* no HIP device / runtime failure at init        */
  BL_ERR_HIP = 2,         /* a HIP call failed (bl_last_error() has text)   */
  BL_ERR_ARG = 3,         /* bad argument (shape, null pointer, P too large) */
  BL_ERR_SAMPLER = 4,     /* a sampler hit an iteration cap / bad shape; the
                             low bits of bl_last_sampler_flags() say which   */
  BL_ERR_NOT_PD = 5       /* posterior precision not positive definite       */
};
/* sampler flag bits (device side ORs them into one word per call) */
enum { BL_ST_ITER_CAP = 1, BL_ST_BAD_SHAPE = 2, BL_ST_ALT_FALLTHROUGH = 4,
       BL_ST_NOT_PD = 8 /* a Cholesky factorisation of the posterior precision failed: reported as BL_ERR_NOT_PD, fatal
                           to the chain (the reference throws there and gibbs() aborts, LogitWrapper.cpp:226-229) */ };

const char *bl_last_error(void);
int         bl_last_sampler_flags(void);
int         bl_device_count(void);
/* select the HIP device this process uses (one process per GPU) */
int         bl_set_device(int device);

/* --------------------------------------------------- Part 1: .C boundary ---
 * The reference brackets each of these with GetRNGstate()/PutRNGstate() and so
 * consumes R's global stream (LogitWrapper.cpp:44-46,59-61).  Here the stream is
 * Philox4x32-10 keyed by a process-global seed; every call below uses the next
 * `epoch` of that seed, so successive calls give fresh draws and a fixed seed
 * reproduces the whole call sequence.  An R-side shim sets the seed from
 * unif_rand() inside its own Get/PutRNGstate bracket. */
void     bl_set_seed(uint64_t seed);      /* also resets the call epoch to 0 */
uint64_t bl_get_seed(void);
/* .C-callable seeding for an R shim: u[0], u[1] are two unif_rand() values in [0,1); the
 * seed is floor(u[0] 2^32) << 32 | floor(u[1] 2^32).  Resets the call epoch like bl_set_seed. */
void     bl_set_seed_from_unif(double *u);
uint32_t bl_get_epoch(void);
/* beta draw used by gibbs(): 1 = the fork's active sign-constrained coordinate
 * draw (Logit.hpp:322-400, call site :429), 0 = unconstrained MVN (Logit.hpp:291-320).
 * Default 1, i.e. what the reference's gibbs() actually executes. */
void     bl_set_constrain(int constrain);
/* the same two knobs with R's .C calling convention (every argument a pointer) */
void     bl_set_constrain_R(int *constrain);
void     bl_set_device_R(int *device, int *rc);

/* LogitWrapper.h:27 */ void rpg_gamma  (double *x, double *n, double *z, int *num, int *trunc);
/* LogitWrapper.h:29 */ void rpg_devroye(double *x, int *n, double *z, int *num);
/* LogitWrapper.h:31 */ void rpg_alt    (double *x, double *h, double *z, int *num);
/* LogitWrapper.h:33 */ void rpg_sp     (double *x, double *h, double *z, int *num, int *iter);
/* LogitWrapper.h:35 */ void rpg_hybrid (double *x, double *h, double *z, int *num);

/* LogitWrapper.h:39-43 */
void gibbs(double *wp, double *betap,
           double *yp, double *tXp, double *np,
           double *m0p, double *P0p,
           int *N, int *P,
           int *samp, int *burn);
/* LogitWrapper.h:45-48 */
void EM(double *betap,
        double *yp, double *tXp, double *np,
        int *Np, int *Pp,
        double *tolp, int *max_iterp);
/* LogitWrapper.h:50-51 */
void combine(double *yp, double *tXp, double *np, int *N, int *P);
/* LogitWrapper.h:55-59 */
void mult_gibbs(double *wp, double *betap,
                double *typ, double *tXp, double *np,
                double *m0p, double *P0p,
                int *N, int *P, int *J,
                int *sampp, int *burnp);
/* LogitWrapper.h:61-62 */
void mult_combine(double *typ, double *tXp, double *np, int *N, int *P, int *J);

/* ------------------------------------------- Part 2: device-resident form ---
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 * launches are asynchronous on that stream; the status word is read back by
 * bl_sync_status(stream).  idx0 is the global index of element 0 (shard offset):
 * element i reads stream (seed, idx0+i, domain, epoch), so a range split over
 * G GPUs yields exactly the draws of the unsplit call. */
int bl_sync_status(void *stream);

/* rpg_devroye (LogitWrapper.cpp:66-85).  n_vec may be NULL: every element then
 * uses n_scalar. */
int bl_rpg_devroye_dev(double *x, const int *n_vec, int n_scalar, const double *z, int64_t num,
                       uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
/* rpg_hybrid (LogitWrapper.cpp:129-167) */
int bl_rpg_hybrid_dev(double *x, const double *h, const double *z, int64_t num,
                      uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
/* rpg_alt / rpg_sp / rpg_gamma (LogitWrapper.cpp:87-106 / 108-127 / 39-62); iter may be NULL */
int bl_rpg_alt_dev  (double *x, const double *h, const double *z, int64_t num,
                     uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
int bl_rpg_sp_dev   (double *x, const double *h, const double *z, int64_t num, int *iter,
                     uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
int bl_rpg_gamma_dev(double *x, const double *h, const double *z, int64_t num, int trunc,
                     uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);

/* synthetic-data helpers (benchmark inputs are generated on the device):
 * out[i] = lo + (hi-lo) U_i ; out[i] = mean + sd N_i ; from (seed, idx0+i, DOM_DATA, epoch) */
int bl_fill_unif_dev(double *out, int64_t num, double lo, double hi,
                     uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
int bl_fill_norm_dev(double *out, int64_t num, double mean, double sd,
                     uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
/* h[i] = 1 + (Philox word mod kmax) : integer shapes 1..kmax as doubles */
int bl_fill_shape_dev(double *out, int64_t num, int kmax,
                      uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
/* y[i] ~ Bernoulli(sigmoid(x_i . beta)) for a P x N column-major tX */
int bl_fill_logit_y_dev(double *y, const double *tX, const double *beta, int64_t N, int P,
                        uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);

/* ---- logistic Gibbs, sharded by observation (Logit.hpp:402-481) ----
 * One handle per process/GPU holds this rank's rows [idx0, idx0+N_local) of the
 * design matrix.  Per sweep s:
 *   bl_gibbs_sweep_local : psi = X beta, omega_i ~ PG(n_i, psi_i), partial
 *                          PP_k = X_k' Omega_k X_k into the handle's P*P buffer
 *   (multi-GPU: the caller all-reduces bl_gibbs_pp_ptr() -- P*P doubles -- over RCCL)
 *   bl_gibbs_draw_beta   : PP += P0, Cholesky, beta draw (redundant on every rank,
 *                          same (seed, sweep) stream => identical beta)
 * bP = P0 m0 + X' kappa is formed once: bl_gibbs_set_bp_local gives this rank's
 * X_k' kappa_k in bl_gibbs_bp_ptr() (all-reduce it once), bl_gibbs_finish_bp adds P0 m0.
 */
typedef struct bl_gibbs bl_gibbs;
int  bl_gibbs_create(bl_gibbs **h, int64_t N_local, int P, uint64_t idx0, uint64_t seed, void *stream);
void bl_gibbs_destroy(bl_gibbs *h);
/* device pointers, not copied: tX (P x N_local col-major), y, n (N_local) must outlive the handle */
int  bl_gibbs_set_data(bl_gibbs *h, const double *tX, const double *y, const double *n);
/* host pointers, copied */
int  bl_gibbs_set_prior(bl_gibbs *h, const double *m0_host, const double *P0_host);
int  bl_gibbs_set_beta(bl_gibbs *h, const double *beta_host);
int  bl_gibbs_set_bp_local(bl_gibbs *h);
int  bl_gibbs_finish_bp(bl_gibbs *h);
/* where a chain driven through the step API begins (bl_gibbs_run / bl_gibbs_run_stream call it themselves): the handle
 * forgets the previous chain's sweep count, deferred-row count (the P = 64 single-pass fall-back decision) and failed-
 * Cholesky flag, so the same seed on the same handle reproduces the same bits */
int  bl_gibbs_chain_start(bl_gibbs *h);
int  bl_gibbs_sweep_local(bl_gibbs *h, uint32_t sweep, double *w_out /* device, N_local, or NULL */);
int  bl_gibbs_draw_beta(bl_gibbs *h, uint32_t sweep, int constrain);
/* EM pieces on the same handle (Logit.hpp:488-554): deterministic omega, then solve */
int  bl_gibbs_em_local(bl_gibbs *h);
int  bl_gibbs_em_solve(bl_gibbs *h, double *dist_host);
double *bl_gibbs_pp_ptr(bl_gibbs *h);     /* device, P*P doubles   */
double *bl_gibbs_bp_ptr(bl_gibbs *h);     /* device, P doubles     */
double *bl_gibbs_beta_ptr(bl_gibbs *h);   /* device, P doubles     */
int  bl_gibbs_get_beta(bl_gibbs *h, double *beta_host);
/* whole single-GPU chain, device-resident data: runs burn + samp sweeps with the
 * reference's slot semantics; beta_out (host, P x samp col-major), w_out (device,
 * N x samp, or NULL = omega not stored). */
int  bl_gibbs_run(bl_gibbs *h, int samp, int burn, int constrain,
                  double *beta_out_host, double *w_out_dev);

/* The same chain with the outputs streamed, thinned or reduced on the device, for sizes where the
 * N x samp array of Logit.hpp:434-444 / LogitWrapper.R:223,239-241 cannot be materialised.
 *   w_mode BL_W_NONE : omega is never stored (w_out ignored)
 *          BL_W_LAST : w_out is a DEVICE buffer of N doubles holding the last sweep's omega
 *          BL_W_ALL  : w_out is a HOST buffer, N x samp column-major, filled sweep by sweep through a
 *                      small device ring while the next sweep runs (device memory O(N), not O(N samp))
 *   thin >= 1        : beta_out_host receives every thin-th sample: P x ceil(samp / thin)
 *   stats (or NULL)  : moments over all samp sweeps, accumulated on the device (Welford):
 *                      beta_mean/beta_var HOST buffers of P doubles (or NULL);
 *                      w_mean/w_var DEVICE buffers of N doubles (or NULL, both or neither).
 * Draws are those of bl_gibbs_run for the same handle seed. */
enum { BL_W_NONE = 0, BL_W_LAST = 1, BL_W_ALL = 2 };
typedef struct {
  double *beta_mean_host, *beta_var_host;
  double *w_mean_dev, *w_var_dev;
} bl_gibbs_stats;
int  bl_gibbs_run_stream(bl_gibbs *h, int samp, int burn, int constrain, int thin,
                         double *beta_out_host, int w_mode, double *w_out, const bl_gibbs_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
