/* bl_oracle.h -- CPU oracle for the Polya-Gamma / logistic-Gibbs hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load liboracle.so.  The product (bayeslogit_amd/csrc) has its own,
 * independent HIP implementation and never links or includes this directory.
 *
 * What it is: a plain-C restatement of the reference algorithms, each
 * function citing the reference file:line it follows (paths relative to the
 * reference checkout, e.g. Code/C/PolyaGamma.cpp:151-202).
 *
 * Pinning status (see oracle/README.md):
 *   - The reference's samplers and Gibbs driver need two third-party headers
 *     that are not in its tree (jwindle/RNG "RNG.hpp", jwindle/Matrix;
 *     INSTALL:14-33): UNBUILDABLE here without stand-ins, which we do not
 *     write.  The one file of the path that needs neither, Code/C/InvertY.cpp
 *     (y_eval, ydy_eval, fdf_eval, v_eval: the saddle-point sampler's
 *     inversion), is compiled UNCHANGED into oracle/_ref/libinverty_ref.so
 *     (`make -C oracle ref`); its outputs on a grid are the golden file
 *     tests/golden/inverty_ref.json, and bl_y_eval / bl_ydy_eval / bl_fdf_eval
 *     / bl_v_eval equal it bit for bit (tests/test_inverty_ref.py).
 *   - The reference holds no golden vectors (no test asserts anything).  The
 *     oracle is pinned by what the reference's own tests/data do hold:
 *       * Code/R/t1to4.txt == trunc_schedule (data file, tests/golden/),
 *       * ygrid/vgrid identity y = tan(sqrt v)/sqrt v (InvertY.hpp:21-57),
 *       * p,q constants of Code/R/lambda-jacobi.R:7-8 -> mass_texpon(0),
 *       * closed-form moments pg_m1/pg_m2 -- the criterion test_pgomp.cpp:56-62
 *         and test_hybrid_par.cpp:55-59 print next to sample moments,
 *       * the exact PG(1,z) CDF series of Code/R/PG.R:320-348.
 *   - PARITY UNPINNED for: the exact random stream (the reference consumes
 *     R's/GSL's sequential stream through the absent RNG library; we use a
 *     Philox4x32-10 counter stream), and the internal algorithms of the
 *     absent library's samplers (gamma, igauss, ltgamma, rtinvchi2, tnorm),
 *     which are restated from the in-tree R prototypes.
 */
#ifndef BL_ORACLE_H
#define BL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- counter RNG stream (contract shared, by specification, with the HIP
 *      product; see DESIGN.md "RNG stream contract") ---- */
typedef struct {
  uint32_t key[2];
  uint32_t ctr[4];   /* idx_lo, idx_hi|domain<<24, epoch, block */
  uint32_t buf[4];
  int      pos;      /* next unread uniform in buf: 0,1 ; 2 = empty */
  uint64_t nunif;    /* uniforms consumed (diagnostic) */
} bl_rng;

enum { BL_DOM_DRAW = 0, BL_DOM_BETA = 1, BL_DOM_DATA = 2, BL_DOM_OMEGA = 3, BL_DOM_KEY = 4 };
/* key of the chain started by the call-th gibbs()/mult_gibbs() of the .C boundary after set_seed(seed) */
uint64_t bl_chain_key(uint64_t seed, uint32_t call);

void   bl_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   bl_rng_init(bl_rng *r, uint64_t seed, uint64_t idx, uint32_t domain, uint32_t epoch);
double bl_unif(bl_rng *r);
double bl_expon_rate(bl_rng *r, double rate);
double bl_norm(bl_rng *r, double mean, double sd);
double bl_gamma_scale(bl_rng *r, double shape, double scale);
double bl_igauss(bl_rng *r, double mu, double lambda);
double bl_ltgamma(bl_rng *r, double shape, double rate, double trunc);
double bl_rtinvchi2(bl_rng *r, double scale, double trunc);
double bl_tnorm(bl_rng *r, double lo, double hi);   /* always 9 uniforms */
double bl_qnorm(double p);
double bl_flat(bl_rng *r, double a, double b);

/* ---- special functions of the absent RNG library ---- */
double bl_p_norm(double x, int use_log);
double bl_p_gamma_rate(double x, double shape, double rate);
double bl_p_igauss(double x, double mu, double lambda);

/* ---- PG samplers ---- */
double bl_pg_a(int n, double x);
double bl_pg_mass_texpon(double Z);
double bl_pg_rtigauss(double Z, bl_rng *r);
typedef struct { double Z, fz, mass, im, il; } bl_pg1_par;
void   bl_pg1_par_of(bl_pg1_par *p, double z);
int    bl_pg1_attempt(int *fresh, double *X, const bl_pg1_par *p, double u1, double u2);
double bl_pg_draw_like_devroye(double z, bl_rng *r);           /* attempt form (= HIP path) */
double bl_pg_draw_devroye(int n, double z, bl_rng *r);
double bl_pg_draw_like_devroye_literal(double z, bl_rng *r);   /* the reference loops, literally */
double bl_pg_draw_devroye_literal(int n, double z, bl_rng *r);
void   bl_pg_devroye_literal_census(double Z, int64_t ndraws, uint64_t seed, int64_t counts[4], int64_t *nprop);
double bl_pg_draw_sum_of_gammas(double b, double z, int trunc, bl_rng *r);
double bl_pg_m1(double b, double z);
double bl_pg_m2(double b, double z);
double bl_jj_m1(double b, double z);
double bl_jj_m2(double b, double z);

extern const double bl_trunc_schedule[301];
double bl_alt_a_coef(int n, double x, double h);
double bl_alt_g_tilde(double x, double h, double trunc);
double bl_alt_w_left(double trunc, double h, double z);
double bl_alt_w_right(double trunc, double h, double z);
double bl_alt_draw_abridged(double h, double z, bl_rng *r, int max_inner);
double bl_alt_draw(double h, double z, bl_rng *r);

extern const double bl_ygrid[81];
extern const double bl_vgrid[81];
double bl_y_eval(double v);
void bl_ydy_eval(double v, double *yp, double *dyp);
void bl_fdf_eval(double v, double y, double *fp, double *dfp);
double bl_v_eval(double y);
double bl_sp_y_func(double v);
double bl_sp_approx(double x, double n, double z);
void   bl_sp_tangent_to_eta(double x, double z, double mid, double *slope, double *icept);
int    bl_sp_draw(double *d, double n, double z, bl_rng *r, int maxiter);

double bl_pg_hybrid(double b, double z, bl_rng *r);

/* ---- the Alt and SP samplers in the attempt form the HIP kernels execute (pg_attempt.c) ---- */
typedef struct { double h, Z, t, fz, p, ip, iq, R, b, ic0, omc, log_m, cR; int small; } bl_alt_par;
void   bl_alt_par_of(bl_alt_par *p, double h, double z);
int    bl_alt_attempt(int *state, double *X, const bl_alt_par *p, double u1, double u2);
double bl_alt_draw_attempt(double h, double z, bl_rng *r);
typedef struct { double n, Z2, md, logmd, lcZ, lhal, lhar, rl, il, rr, ir, mu, pl, ipl, iql, b, ic0, omc, log_m; } bl_sp_par;
extern const double bl_vtab[3][16][11];
void   bl_sp_vlk(double x, double logx, double *v, double *L, double *lK2);
void   bl_sp_par_of(bl_sp_par *p, double n, double z);
int    bl_sp_attempt(int *state, double *X, int *accepted, const bl_sp_par *p, double u1, double u2);
int    bl_sp_draw_attempt(double *d, double n, double z, bl_rng *r, int maxiter);
double bl_upper_gamma_cf(double a, double x);
double bl_pg_hybrid_attempt(double b, double z, bl_rng *r);

/* ---- vector entry points mirroring Code/C/LogitWrapper.h:23-64, plus the
 *      (seed, epoch, index offset) of the counter stream ---- */
void bl_o_rpg_devroye(double *x, const int *n, const double *z, int64_t num,
                      uint64_t seed, uint32_t epoch, uint64_t idx0);
void bl_o_rpg_alt    (double *x, const double *h, const double *z, int64_t num,
                      uint64_t seed, uint32_t epoch, uint64_t idx0);
void bl_o_rpg_sp     (double *x, const double *h, const double *z, int64_t num, int *iter,
                      uint64_t seed, uint32_t epoch, uint64_t idx0);
void bl_o_rpg_gamma  (double *x, const double *h, const double *z, int64_t num, int trunc,
                      uint64_t seed, uint32_t epoch, uint64_t idx0);
void bl_o_rpg_hybrid (double *x, const double *h, const double *z, int64_t num,
                      uint64_t seed, uint32_t epoch, uint64_t idx0);
/* the same three on the attempt forms (what the HIP path computes, draw for draw); nblk (may be
 * NULL) receives the number of Philox blocks each observation consumed */
void bl_o_rpg_alt_attempt   (double *x, const double *h, const double *z, int64_t num,
                             uint64_t seed, uint32_t epoch, uint64_t idx0, uint32_t *nblk);
void bl_o_rpg_sp_attempt    (double *x, const double *h, const double *z, int64_t num, int *iter,
                             uint64_t seed, uint32_t epoch, uint64_t idx0, uint32_t *nblk);
void bl_o_rpg_hybrid_attempt(double *x, const double *h, const double *z, int64_t num,
                             uint64_t seed, uint32_t epoch, uint64_t idx0);
/* OpenMP variant of the hybrid loop, one stream per observation, schedule(dynamic)
 * as Code/C/PolyaGammaOMP.h:61-71 (timed CPU baseline on all host cores). */
void bl_o_rpg_hybrid_omp(double *x, const double *h, const double *z, int64_t num,
                         uint64_t seed, uint32_t epoch, uint64_t idx0, int nthreads);
void bl_o_rpg_devroye_omp(double *x, const int *n, const double *z, int64_t num,
                          uint64_t seed, uint32_t epoch, uint64_t idx0, int nthreads, int literal);
int  bl_o_max_threads(void);

/* ---- Gibbs / EM / combine (Code/C/Logit.hpp, MultLogit.hpp) ---- */
/* constrain: 1 = the fork's active truncated-normal coordinate draw
 * (Logit.hpp:322-400), 0 = the unconstrained MVN draw (Logit.hpp:291-320).
 * w may be NULL (omega not stored).  idx0 = global index of local row 0. */
int bl_o_gibbs(double *w, double *beta, const double *y, const double *tX, const double *n,
               const double *m0, const double *P0, int64_t N, int P, int samp, int burn,
               uint64_t seed, int constrain, uint64_t idx0);
/* one sweep's shard-local pieces, used by the multi-rank host-logic tests */
void bl_o_sweep_partial(double *PPpart, double *w, const double *tX, const double *n,
                        const double *beta, int64_t N, int P,
                        uint64_t seed, uint32_t sweep, uint64_t idx0);
void bl_o_draw_beta(double *beta, const double *PP, const double *bP, const double *beta_prev,
                    int P, uint64_t seed, uint32_t sweep, int constrain);
void bl_o_set_bP(double *bP, const double *y, const double *tX, const double *n,
                 const double *m0, const double *P0, int64_t N, int P);
int  bl_o_EM(double *beta, const double *y, const double *tX, const double *n,
             int64_t N, int P, double tol, int max_iter);
int64_t bl_o_combine(double *y, double *tX, double *n, int64_t N, int P);
int64_t bl_o_mult_combine(double *ty, double *tX, double *n, int64_t N, int P, int J);
int bl_o_mult_gibbs(double *w, double *beta, const double *ty, const double *tX, const double *n,
                    const double *m0, const double *P0, int64_t N, int P, int J,
                    int samp, int burn, uint64_t seed);

/* small dense linear algebra (the absent Matrix library's BLAS/LAPACK calls) */
int  bl_chol_upper(double *U, const double *A, int P);  /* A = U'U, column-major */
int  bl_chol_lower(double *L, const double *A, int P);  /* A = L L' */

#ifdef __cplusplus
}
#endif
#endif
