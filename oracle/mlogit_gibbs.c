/* mlogit_gibbs.c -- oracle (test infrastructure, see bl_oracle.h).
 * Restates Code/C/MultLogit.hpp (multinomial logit Gibbs) and
 * Code/C/include/Normal.hpp:98-131 (N(m,V) from likelihood form), plain loops.
 * Layouts (LogitWrapper.cpp:325-330): ty (J-1) x N, tX P x N, m0 P x (J-1),
 * P0 P x P x (J-1), w N x (J-1) x samp, beta P x (J-1) x samp; column-major.
 *
 * RNG streams: sweep s, category j use epoch e = s*(J-1)+j: omega_i from
 * (seed, i, DOM_OMEGA, e), beta_j from (seed, 0, DOM_BETA, e).
 */
#include "bl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define A_(M, i, j, ld) ((M)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

/* MultLogit::set_data merge -- MultLogit.hpp:137-208.  In place, returns new N. */
int64_t bl_o_mult_combine(double *ty, double *tX, double *n, int64_t N, int P, int J)
{
  int U = J - 1;
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
        for (int k = 0; k < U; ++k)
          ty[(size_t)i * U + k] = (n[i] / sum) * ty[(size_t)i * U + k] + (n[j] / sum) * ty[(size_t)j * U + k];
        n[i] = sum;
        dead[j] = 1;
      }
    }
  }
  int64_t M = 0;
  for (int64_t i = 0; i < N; ++i) {
    if (dead[i]) continue;
    if (M != i) {
      n[M] = n[i];
      memmove(ty + (size_t)M * U, ty + (size_t)i * U, sizeof(double) * U);
      memmove(tX + (size_t)M * P, tX + (size_t)i * P, sizeof(double) * P);
    }
    ++M;
  }
  free(dead);
  return M;
}

/* Normal::set_from_likelihood(b, R) + Normal::draw -- Normal.hpp:98-131:
 * V = R^{-1}; mean = V b; lower = chol(V,'L'); d = mean + lower * eps. */
static void mvn_from_likelihood_draw(double *out, const double *b, const double *R, int P, bl_rng *r)
{
  size_t sz = sizeof(double) * P * P;
  double *U = (double *)malloc(sz), *V = (double *)malloc(sz), *Lw = (double *)malloc(sz);
  double *eps = (double *)malloc(sizeof(double) * P);
  bl_chol_upper(U, R, P);
  for (int j = 0; j < P; ++j) {                 /* symsolve(R, V = I) */
    double *col = V + (size_t)j * P;
    for (int i = 0; i < P; ++i) col[i] = (i == j) ? 1.0 : 0.0;
    for (int i = 0; i < P; ++i) {               /* U' y = e_j */
      double s = col[i];
      for (int k = 0; k < i; ++k) s -= A_(U, k, i, P) * col[k];
      col[i] = s / A_(U, i, i, P);
    }
    for (int i = P - 1; i >= 0; --i) {          /* U x = y */
      double s = col[i];
      for (int k = i + 1; k < P; ++k) s -= A_(U, i, k, P) * col[k];
      col[i] = s / A_(U, i, i, P);
    }
  }
  bl_chol_lower(Lw, V, P);
  for (int i = 0; i < P; ++i) eps[i] = bl_norm(r, 0.0, 1.0);   /* r.norm(d, 0, 1), :123 */
  for (int i = 0; i < P; ++i) {
    double mean = 0.0, le = 0.0;
    for (int k = 0; k < P; ++k) mean += A_(V, i, k, P) * b[k];  /* gemm(mean, V, b), :107 */
    for (int k = 0; k <= i; ++k) le += A_(Lw, i, k, P) * eps[k]; /* trmm(lower, d, 'L'), :126 */
    out[i] = le + mean;
  }
  free(U); free(V); free(Lw); free(eps);
}

/* MultLogit::gibbs -- MultLogit.hpp:261-372 (data already merged by the
 * caller, as mlogit() does through mlogit.combine, LogitWrapper.R:371-377). */
int bl_o_mult_gibbs(double *w, double *beta, const double *ty, const double *tX, const double *n,
                    const double *m0, const double *P0, int64_t N, int P, int J,
                    int samp, int burn, uint64_t seed)
{
  int U = J - 1;
  if (samp < 1 || burn < 0 || U < 1) return -1;
  size_t PP = (size_t)P * P;
  double *Z = (double *)calloc((size_t)P * U, sizeof(double));
  double *b0 = (double *)calloc((size_t)P * U, sizeof(double));
  double *XB = (double *)calloc((size_t)N * J, sizeof(double));      /* last column stays 0 */
  double *XBnoj = (double *)malloc(sizeof(double) * (size_t)N * U);
  double *cj = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  double *eta = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
  double *P1 = (double *)malloc(sizeof(double) * PP);
  double *b1 = (double *)malloc(sizeof(double) * P);
  double *bnew = (double *)malloc(sizeof(double) * P);
  double *wscr = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));

  /* Z = tX tkappa', tkappa(:,i) = n_i (ty(:,i) - 1/2) -- :214-219 */
  for (int64_t i = 0; i < N; ++i)
    for (int j = 0; j < U; ++j) {
      double k = n[i] * (ty[(size_t)i * U + j] - 0.5);
      for (int p = 0; p < P; ++p) Z[(size_t)j * P + p] += tX[(size_t)i * P + p] * k;
    }
  /* b0_j = P0_j m0_j -- :272-273 */
  for (int j = 0; j < U; ++j)
    for (int a = 0; a < P; ++a) {
      double s = 0.0;
      for (int b = 0; b < P; ++b) s += P0[(size_t)j * PP + a + (size_t)b * P] * m0[(size_t)j * P + b];
      b0[(size_t)j * P + a] = s;
    }
  memset(beta, 0, sizeof(double) * (size_t)P * U * samp);
  if (w) memset(w, 0, sizeof(double) * (size_t)N * U * samp);

  int total = burn + samp;          /* burn+1 sweeps into slot 0, then samp-1 more: :284,:332 */
  for (int s = 0; s < total; ++s) {
    int slot = s <= burn ? 0 : s - burn;
    int prev = s <= burn ? 0 : slot - 1;
    double *bslot = beta + (size_t)slot * P * U;
    const double *bprev = beta + (size_t)prev * P * U;
    (void)bprev;                    /* beta_prev is passed but unused by MultLogit::draw_beta, :242-258 */
    /* XB_no_j = XB[, 1..J-1] -- :286,:334 */
    for (int c = 0; c < U; ++c)
      memcpy(XBnoj + (size_t)c * N, XB + (size_t)(c + 1) * N, sizeof(double) * N);
    for (int j = 0; j < U; ++j) {
      uint32_t epoch = (uint32_t)s * (uint32_t)U + (uint32_t)j;
      for (int64_t i = 0; i < N; ++i) {          /* A = rowSums(exp(XB_no_j)); c_j = log A; eta = XB_j - c_j */
        double A = 0.0;
        for (int c = 0; c < U; ++c) A += exp(XBnoj[(size_t)c * N + i]);
        cj[i] = log(A);
        eta[i] = XB[(size_t)j * N + i] - cj[i];
      }
      double *wj = w ? w + (size_t)slot * N * U + (size_t)j * N : wscr;
      for (int64_t i = 0; i < N; ++i) {          /* draw_w -- :234-240 */
        bl_rng r;
        bl_rng_init(&r, seed, (uint64_t)i, BL_DOM_OMEGA, epoch);
        wj[i] = bl_pg_draw_devroye((int)n[i], eta[i], &r);
      }
      /* draw_beta -- :242-258 */
      memcpy(P1, P0 + (size_t)j * PP, sizeof(double) * PP);
      for (int p = 0; p < P; ++p) b1[p] = Z[(size_t)j * P + p] + b0[(size_t)j * P + p];
      for (int64_t i = 0; i < N; ++i) {
        const double *x = tX + (size_t)i * P;
        double wi = wj[i];
        for (int b = 0; b < P; ++b) {
          double xw = x[b] * wi;
          b1[b] += xw * cj[i];
          for (int a = 0; a < P; ++a) A_(P1, a, b, P) += x[a] * xw;
        }
      }
      bl_rng rb;
      bl_rng_init(&rb, seed, 0, BL_DOM_BETA, epoch);
      mvn_from_likelihood_draw(bnew, b1, P1, P, &rb);
      memcpy(bslot + (size_t)j * P, bnew, sizeof(double) * P);
      for (int64_t i = 0; i < N; ++i) {          /* gemm(XB.col(j), tX, beta_j, 'T') -- :307 */
        const double *x = tX + (size_t)i * P;
        double sdot = 0.0;
        for (int p = 0; p < P; ++p) sdot += x[p] * bnew[p];
        XB[(size_t)j * N + i] = sdot;
      }
      if (j < U - 1)                             /* :313 */
        memcpy(XBnoj + (size_t)j * N, XB + (size_t)j * N, sizeof(double) * N);
    }
    /* a new slot starts from the previous slot's other categories only through XB */
  }
  free(Z); free(b0); free(XB); free(XBnoj); free(cj); free(eta); free(P1); free(b1); free(bnew); free(wscr);
  return 0;
}
