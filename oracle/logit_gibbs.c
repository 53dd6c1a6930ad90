/* logit_gibbs.c -- oracle (test infrastructure, see bl_oracle.h).
 * Restates Code/C/Logit.hpp (binomial logit Gibbs, EM, data merge) with plain
 * loops standing in for the absent Matrix library's BLAS/LAPACK wrappers.
 * All matrices column-major; tX is P x N (observation i = column i).
 *
 * RNG streams (DESIGN.md "RNG stream contract"): sweep s (burn-in sweeps
 * 0..burn-1, sampling sweeps burn..burn+samp-1) draws omega_i from
 * (seed, idx0+i, DOM_OMEGA, epoch=s) and beta from (seed, 0, DOM_BETA, epoch=s).
 */
#include "bl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define A_(M, i, j, ld) ((M)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

/* chol(U, A, 'U'): A = U'U */
int bl_chol_upper(double *U, const double *A, int P)
{
  memset(U, 0, sizeof(double) * P * P);
  for (int j = 0; j < P; ++j) {
    for (int i = 0; i <= j; ++i) {
      double s = A_(A, i, j, P);
      for (int k = 0; k < i; ++k) s -= A_(U, k, i, P) * A_(U, k, j, P);
      if (i == j) {
        if (!(s > 0.0)) return j + 1;
        A_(U, j, j, P) = sqrt(s);
      } else {
        A_(U, i, j, P) = s / A_(U, i, i, P);
      }
    }
  }
  return 0;
}

/* chol(L, A, 'L'): A = L L' */
int bl_chol_lower(double *L, const double *A, int P)
{
  memset(L, 0, sizeof(double) * P * P);
  for (int j = 0; j < P; ++j) {
    double s = A_(A, j, j, P);
    for (int k = 0; k < j; ++k) s -= A_(L, j, k, P) * A_(L, j, k, P);
    if (!(s > 0.0)) return j + 1;
    double d = sqrt(s);
    A_(L, j, j, P) = d;
    for (int i = j + 1; i < P; ++i) {
      double t = A_(A, i, j, P);
      for (int k = 0; k < j; ++k) t -= A_(L, i, k, P) * A_(L, j, k, P);
      A_(L, i, j, P) = t / d;
    }
  }
  return 0;
}

/* trsm(U, b, 'U','L','T'): solve U' y = b in place (forward) */
static void solve_Ut(const double *U, double *b, int P)
{
  for (int i = 0; i < P; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A_(U, k, i, P) * b[k];
    b[i] = s / A_(U, i, i, P);
  }
}
/* trsm(U, b, 'U','L','N'): solve U x = b in place (backward) */
static void solve_U(const double *U, double *b, int P)
{
  for (int i = P - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < P; ++k) s -= A_(U, i, k, P) * b[k];
    b[i] = s / A_(U, i, i, P);
  }
}
/* trsm(L, b, 'L','L','N'): solve L x = b in place (forward) */
static void solve_L(const double *L, double *b, int P)
{
  for (int i = 0; i < P; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A_(L, i, k, P) * b[k];
    b[i] = s / A_(L, i, i, P);
  }
}

/* Logit::set_prior + Logit::set_bP -- Logit.hpp:185-190,174-183:
 * b0 = P0 m0; alpha_i = n_i (y_i - 1/2); bP = b0 + tX alpha. */
void bl_o_set_bP(double *bP, const double *y, const double *tX, const double *n,
                 const double *m0, const double *P0, int64_t N, int P)
{
  for (int i = 0; i < P; ++i) {
    double s = 0.0;
    for (int j = 0; j < P; ++j) s += A_(P0, i, j, P) * m0[j];
    bP[i] = s;
  }
  for (int64_t i = 0; i < N; ++i) {
    double alpha = n[i] * (y[i] - 0.5);
    const double *x = tX + (size_t)i * P;
    for (int j = 0; j < P; ++j) bP[j] += x[j] * alpha;
  }
}

/* psi = tX' beta -- gemm(psi, tX, beta, 'T'), Logit.hpp:421,431 */
static void calc_psi(double *psi, const double *tX, const double *beta, int64_t N, int P)
{
  for (int64_t i = 0; i < N; ++i) {
    const double *x = tX + (size_t)i * P;
    double s = 0.0;
    for (int j = 0; j < P; ++j) s += x[j] * beta[j];
    psi[i] = s;
  }
}

/* Logit::draw_w -- Logit.hpp:283-289: w_i = pg.draw((int) n_i, psi_i, r) */
static void draw_w(double *w, const double *psi, const double *n, int64_t N,
                   uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  for (int64_t i = 0; i < N; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_OMEGA, epoch);
    w[i] = bl_pg_draw_devroye((int)n[i], psi[i], &r);
  }
}

/* tXRtOm = tX sqrt(Om); syrk(PP, tXRtOm) -- Logit.hpp:294-301 / :325-332.
 * PPpart (+)= sum_i (x_i sqrt w_i)(x_i sqrt w_i)' ; full symmetric matrix. */
static void accum_xwx(double *PP, const double *tX, const double *w, int64_t N, int P)
{
  double *xs = (double *)malloc(sizeof(double) * P);
  for (int64_t i = 0; i < N; ++i) {
    const double *x = tX + (size_t)i * P;
    double rt = sqrt(w[i]);
    for (int j = 0; j < P; ++j) xs[j] = x[j] * rt;
    for (int b = 0; b < P; ++b)
      for (int a = 0; a <= b; ++a)
        A_(PP, a, b, P) += xs[a] * xs[b];
  }
  for (int b = 0; b < P; ++b)
    for (int a = 0; a < b; ++a)
      A_(PP, b, a, P) = A_(PP, a, b, P);
  free(xs);
}

void bl_o_sweep_partial(double *PPpart, double *w, const double *tX, const double *n,
                        const double *beta, int64_t N, int P,
                        uint64_t seed, uint32_t sweep, uint64_t idx0)
{
  double *psi = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  double *wl = w ? w : (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  calc_psi(psi, tX, beta, N, P);
  draw_w(wl, psi, n, N, seed, sweep, idx0);
  memset(PPpart, 0, sizeof(double) * P * P);
  accum_xwx(PPpart, tX, wl, N, P);
  if (!w) free(wl);
  free(psi);
}

/* The two beta draws given the posterior precision PP = P0 + X'Omega X.
 * constrain = 0: Logit::draw_beta(beta, w, r), Logit.hpp:291-320.
 * constrain = 1: Logit::draw_beta(beta, w, beta_prev, r), Logit.hpp:322-400
 *                (the call active in gibbs_block, :429). */
void bl_o_draw_beta(double *beta, const double *PP, const double *bP, const double *beta_prev,
                    int P, uint64_t seed, uint32_t sweep, int constrain)
{
  bl_rng r;
  bl_rng_init(&r, seed, 0, BL_DOM_BETA, sweep);
  size_t PPsz = sizeof(double) * P * P;
  double *U = (double *)malloc(PPsz);
  double *mP = (double *)malloc(sizeof(double) * P);
  bl_chol_upper(U, PP, P);                                   /* :304-305 / :335-336 */

  if (!constrain) {
    for (int i = 0; i < P; ++i) beta[i] = bl_norm(&r, 0.0, 1.0);   /* r.norm(beta, 1.0) :311 */
    memcpy(mP, bP, sizeof(double) * P);
    solve_Ut(U, mP, P);                                      /* :314 */
    solve_U(U, mP, P);                                       /* :315 */
    solve_U(U, beta, P);                                     /* :316 */
    for (int i = 0; i < P; ++i) beta[i] += mP[i];            /* :318-319 */
    free(U); free(mP);
    return;
  }

  double *S = (double *)malloc(PPsz);
  double *L = (double *)malloc(PPsz);
  double *z = (double *)malloc(sizeof(double) * P);
  int *is = (int *)malloc(sizeof(int) * P);
  /* S = PP^{-1}: two triangular solves on the identity, :339-348 */
  for (int j = 0; j < P; ++j) {
    double *col = S + (size_t)j * P;
    for (int i = 0; i < P; ++i) col[i] = (i == j) ? 1.0 : 0.0;
    solve_Ut(U, col, P);
    solve_U(U, col, P);
  }
  bl_chol_lower(L, S, P);                                    /* :350-352 */
  memcpy(mP, bP, sizeof(double) * P);
  solve_Ut(U, mP, P);                                        /* :356 */
  solve_U(U, mP, P);                                         /* :357 */
  for (int i = 0; i < P; ++i) {                              /* :360-365 */
    z[i] = beta_prev[i] - mP[i];
    beta[i] = beta_prev[i];
  }
  solve_L(L, z, P);                                          /* :366 */
  for (int i = 0; i < P; ++i) is[i] = i;                     /* :368-371 */
  const double inf = INFINITY;
  for (int k = 0; k < P; ++k) {                              /* :373 */
    for (int i = 0; i < P - 1; ++i) {                        /* random sweep :375-377 */
      int j = (int)(unsigned)bl_flat(&r, (double)i, (double)P);
      int t = is[i]; is[i] = is[j]; is[j] = t;
    }
    for (int i = 0; i < P; i++) {                            /* :380-398 */
      double cmin = -inf, cmax = inf, c1;
      int c = is[i];
      double l1, z1 = z[c], z2;
      for (int j = c; j < P - 1; j++) {
        l1 = A_(L, j, c, P);
        c1 = z1 - beta[j] / l1;
        if (l1 > 0.0 && c1 > cmin) {
          cmin = c1;
        } else if (l1 < 0.0 && c1 < cmax) {
          cmax = c1;
        }
      }
      z2 = bl_tnorm(&r, cmin, cmax);
      z[c] = z2;
      for (int j = c; j < P; j++)
        beta[j] += A_(L, j, c, P) * (z2 - z1);
    }
  }
  free(S); free(L); free(z); free(is); free(U); free(mP);
}

/* Logit::gibbs + gibbs_block -- Logit.hpp:460-481, 402-457.
 * Slot semantics (:434-444): the burn block (samp=1, period=burn) rewrites
 * slot 0 in place; the sampling block starts from slot 0's state, writes sweep
 * m into slot m-1... i.e. sweep 1 overwrites slot 0 using itself as beta_prev,
 * then beta_prev = slot m-1, beta_curr = slot m.  beta/w start at zero. */
int bl_o_gibbs(double *w, double *beta, const double *y, const double *tX, const double *n,
               const double *m0, const double *P0, int64_t N, int P, int samp, int burn,
               uint64_t seed, int constrain, uint64_t idx0)
{
  if (samp < 1 || burn < 0 || P < 1 || N < 0) return -1;
  size_t PPsz = sizeof(double) * P * P;
  double *bP = (double *)malloc(sizeof(double) * P);
  double *PP = (double *)malloc(PPsz);
  double *psi = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  double *wscratch = w ? NULL : (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  double *bnew = (double *)malloc(sizeof(double) * P);

  bl_o_set_bP(bP, y, tX, n, m0, P0, N, P);                   /* set_bP, :462 */
  memset(beta, 0, sizeof(double) * P * samp);
  if (w) memset(w, 0, sizeof(double) * (size_t)N * samp);

  uint32_t sweep = 0;
  for (int block = 0; block < 2; ++block) {
    int bsamp = block == 0 ? 1 : samp;
    int period = block == 0 ? burn : 1;
    double *beta_curr = beta, *beta_prev = beta;
    double *w_curr = w ? w : wscratch;
    calc_psi(psi, tX, beta_curr, N, P);                      /* :421 */
    for (int m = 1; m <= bsamp * period; m++) {              /* :426 */
      draw_w(w_curr, psi, n, N, seed, sweep, idx0);          /* :428 */
      memcpy(PP, P0, PPsz);
      accum_xwx(PP, tX, w_curr, N, P);
      bl_o_draw_beta(bnew, PP, bP, beta_prev, P, seed, sweep, constrain);   /* :429 */
      memcpy(beta_curr, bnew, sizeof(double) * P);
      calc_psi(psi, tX, beta_curr, N, P);                    /* :431 */
      if (m % period == 0) {                                 /* :434-444 */
        beta_prev = beta_curr;
        beta_curr += P;
        if (w) w_curr += N;
      }
      ++sweep;
    }
  }
  free(bP); free(PP); free(psi); free(bnew);
  if (wscratch) free(wscratch);
  return 0;
}

/* Logit::EM -- Logit.hpp:488-554.  P0 = 0, b0 = 0 (EM() builds Logit without
 * set_prior, LogitWrapper.cpp:238-262). */
int bl_o_EM(double *beta, const double *y, const double *tX, const double *n,
            int64_t N, int P, double tol, int max_iter)
{
  size_t PPsz = sizeof(double) * P * P;
  double *bP = (double *)calloc(P, sizeof(double));
  double *PP = (double *)malloc(PPsz);
  double *U = (double *)malloc(PPsz);
  double *psi = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  double *w = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  double *old = (double *)malloc(sizeof(double) * P);
  double *zeroP = (double *)calloc((size_t)P * P + P, sizeof(double));
  bl_o_set_bP(bP, y, tX, n, zeroP, zeroP + P, N, P);
  for (int i = 0; i < P; ++i) beta[i] = 0.0;
  double dist = tol + 1.0;
  int iter = 0;
  while (dist > tol && iter < max_iter) {
    calc_psi(psi, tX, beta, N, P);                           /* :508 */
    for (int64_t i = 0; i < N; ++i) {                        /* :509-519 */
      double hpsi = psi[i] * 0.5;
      if (fabs(hpsi) < 0.01)
        w[i] = n[i] / cosh(hpsi)
             * (1 + hpsi * hpsi / 6.0 + pow(hpsi, 4.0) / 120.0 + pow(hpsi, 6) / 5040.0) * 0.25;
      else
        w[i] = n[i] * tanh(hpsi) / hpsi * 0.25;
    }
    memcpy(old, beta, sizeof(double) * P);
    memset(PP, 0, PPsz);                                     /* PP = P0 = 0, :533 */
    accum_xwx(PP, tX, w, N, P);                              /* :524-535 */
    if (bl_chol_upper(U, PP, P) != 0) break;                 /* :537 */
    memcpy(beta, bP, sizeof(double) * P);                    /* :538-540 */
    solve_Ut(U, beta, P);
    solve_U(U, beta, P);
    dist = 0.0;                                              /* :546-547 */
    for (int i = 0; i < P; ++i) {
      double d = fabs(beta[i] - old[i]);
      if (d > dist) dist = d;
    }
    ++iter;
  }
  free(bP); free(PP); free(U); free(psi); free(w); free(old); free(zeroP);
  return iter;
}

/* Logit::compress -- Logit.hpp:192-270: merge observations with identical
 * covariate columns, keeping first-occurrence order; O(N^2 P).  In place;
 * returns the new N. */
int64_t bl_o_combine(double *y, double *tX, double *n, int64_t N, int P)
{
  char *dead = (char *)calloc(N > 0 ? N : 1, 1);
  for (int64_t i = 0; i < N; ++i) {
    if (dead[i]) continue;
    for (int64_t j = i + 1; j < N; ++j) {
      if (dead[j]) continue;
      int same = 1;
      for (int k = 0; k < P; ++k)
        if (tX[(size_t)i * P + k] != tX[(size_t)j * P + k]) { same = 0; break; }
      if (same) {
        double sum = n[i] + n[j];
        y[i] = (n[i] / sum) * y[i] + (n[j] / sum) * y[j];
        n[i] = sum;
        dead[j] = 1;
      }
    }
  }
  int64_t M = 0;
  for (int64_t i = 0; i < N; ++i) {
    if (dead[i]) continue;
    if (M != i) {
      y[M] = y[i];
      n[M] = n[i];
      memmove(tX + (size_t)M * P, tX + (size_t)i * P, sizeof(double) * P);
    }
    ++M;
  }
  free(dead);
  return M;
}
