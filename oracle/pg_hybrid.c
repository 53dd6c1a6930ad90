/* pg_hybrid.c -- oracle (test infrastructure, see bl_oracle.h).
 * Restates the vector entry points of Code/C/LogitWrapper.cpp:39-167.
 * The reference runs one sequential RNG over the loop; here observation i of a
 * call reads its own counter stream (seed, idx0+i, DOM_DRAW, epoch), so the
 * result does not depend on loop order, thread count or GPU count.
 */
#include "bl_oracle.h"
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* rpg_hybrid dispatch -- LogitWrapper.cpp:140-162 */
double bl_pg_hybrid(double b, double z, bl_rng *r)
{
  double x;
  if (b > 170) {
    double m = bl_pg_m1(b, z);
    double v = bl_pg_m2(b, z) - m * m;
    x = bl_norm(r, m, sqrt(v));
  } else if (b > 13) {
    bl_sp_draw(&x, b, z, r, 200);
  } else if (b == 1 || b == 2) {
    x = bl_pg_draw_devroye((int)b, z, r);
  } else if (b > 1) {
    x = bl_alt_draw(b, z, r);
  } else if (b > 0) {
    x = bl_pg_draw_sum_of_gammas(b, z, 200, r);   /* PolyaGamma dv; default T=200, PolyaGamma.h:51 */
  } else {
    x = 0.0;
  }
  return x;
}

/* rpg_devroye -- LogitWrapper.cpp:66-85 */
void bl_o_rpg_devroye(double *x, const int *n, const double *z, int64_t num,
                      uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    x[i] = (n[i] != 0) ? bl_pg_draw_devroye(n[i], z[i], &r) : 0.0;
  }
}

/* `literal` != 0: the reference loops call for call (bl_pg_draw_devroye_literal) instead of the
 * attempt form; that is the CPU baseline bench.py reports. */
void bl_o_rpg_devroye_omp(double *x, const int *n, const double *z, int64_t num,
                          uint64_t seed, uint32_t epoch, uint64_t idx0, int nthreads, int literal)
{
  (void)nthreads;
  /* thread-level strategy of Code/C/PolyaGammaOMP.h:61-71: dynamic schedule */
  #pragma omp parallel for schedule(dynamic, 4096) num_threads(nthreads)
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    if (n[i] == 0) x[i] = 0.0;
    else x[i] = literal ? bl_pg_draw_devroye_literal(n[i], z[i], &r) : bl_pg_draw_devroye(n[i], z[i], &r);
  }
}

/* rpg_alt -- LogitWrapper.cpp:87-106 */
void bl_o_rpg_alt(double *x, const double *h, const double *z, int64_t num,
                  uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    x[i] = (h[i] != 0) ? bl_alt_draw(h[i], z[i], &r) : 0.0;
  }
}

/* rpg_sp -- LogitWrapper.cpp:108-127 (iter[i] untouched when h[i]==0) */
void bl_o_rpg_sp(double *x, const double *h, const double *z, int64_t num, int *iter,
                 uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    if (h[i] != 0) {
      int it = bl_sp_draw(&x[i], h[i], z[i], &r, 200);
      if (iter) iter[i] = it;
    } else {
      x[i] = 0.0;
    }
  }
}

/* rpg_gamma -- LogitWrapper.cpp:39-62 */
void bl_o_rpg_gamma(double *x, const double *h, const double *z, int64_t num, int trunc,
                    uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    x[i] = (h[i] != 0.0) ? bl_pg_draw_sum_of_gammas(h[i], z[i], trunc, &r) : 0.0;
  }
}

/* rpg_hybrid -- LogitWrapper.cpp:129-167 */
void bl_o_rpg_hybrid(double *x, const double *h, const double *z, int64_t num,
                     uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    x[i] = bl_pg_hybrid(h[i], z[i], &r);
  }
}

/* the attempt forms of pg_attempt.c, observation by observation on the same streams */
void bl_o_rpg_alt_attempt(double *x, const double *h, const double *z, int64_t num,
                          uint64_t seed, uint32_t epoch, uint64_t idx0, uint32_t *nblk)
{
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    x[i] = (h[i] != 0) ? bl_alt_draw_attempt(h[i], z[i], &r) : 0.0;
    if (nblk) nblk[i] = (uint32_t)(r.nunif / 2);
  }
}

void bl_o_rpg_sp_attempt(double *x, const double *h, const double *z, int64_t num, int *iter,
                         uint64_t seed, uint32_t epoch, uint64_t idx0, uint32_t *nblk)
{
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    if (h[i] != 0) {
      int it = bl_sp_draw_attempt(&x[i], h[i], z[i], &r, 200);
      if (iter) iter[i] = it;
    } else {
      x[i] = 0.0;
    }
    if (nblk) nblk[i] = (uint32_t)(r.nunif / 2);
  }
}

void bl_o_rpg_hybrid_attempt(double *x, const double *h, const double *z, int64_t num,
                             uint64_t seed, uint32_t epoch, uint64_t idx0)
{
  #pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    x[i] = bl_pg_hybrid_attempt(h[i], z[i], &r);
  }
}

void bl_o_rpg_hybrid_omp(double *x, const double *h, const double *z, int64_t num,
                         uint64_t seed, uint32_t epoch, uint64_t idx0, int nthreads)
{
  (void)nthreads;
  #pragma omp parallel for schedule(dynamic, 4096) num_threads(nthreads)
  for (int64_t i = 0; i < num; ++i) {
    bl_rng r;
    bl_rng_init(&r, seed, idx0 + (uint64_t)i, BL_DOM_DRAW, epoch);
    x[i] = bl_pg_hybrid(h[i], z[i], &r);
  }
}

int bl_o_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
