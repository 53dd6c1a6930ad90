/* bayeslogit_hip_diag.h -- diagnostic entry points of libbayeslogit_hip.so.
 *
 * NOT part of the drop-in boundary (include/bayeslogit_hip.h): nothing here replaces a reference interface.
 * These are the knobs and probes the benchmark, the profiling scripts and the parity tests use: measured
 * instruction rates for the roofline objects, A/B switches between bit-identical kernels, counters of the
 * work the samplers did, and a device evaluation of the fitted saddle-point inversion table.
 */
#ifndef BAYESLOGIT_HIP_DIAG_H
#define BAYESLOGIT_HIP_DIAG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic (bench.py's roofline): one launch of a register-only loop of `iters` x 10 independent
 * v_mfma_f64_16x16x4_f64 per wave, `waves_per_simd` (1..8) waves on every SIMD; *flops = the flops that launch
 * performs (the caller times it with events on `stream`).  work: >= CUs * waves_per_simd * 256 doubles. */
int bl_diag_mfma_f64_dev(double *work, int64_t work_doubles, int waves_per_simd, int iters, double *flops,
                         void *stream);
/* the same loop on v_mfma_f64_4x4x4_4b_f64 (512 flops), the instruction the X' Omega X kernels for P >= 64 are built on */
int bl_diag_mfma_f64_small_dev(double *work, int64_t work_doubles, int waves_per_simd, int iters, double *flops,
                               void *stream);
/* The Gibbs sweep over a rank's rows (Logit.hpp:283-301,431: psi = X beta, omega ~ PG(n, psi), X' Omega X) reads X
 * once when P = 64 (single_pass = 1, the default; env BL_SWEEP_SINGLE_PASS) or in two streaming passes
 * (single_pass = 0; every other P).  Same omega bit for bit; X' Omega X in another, equally fixed, summation order.
 * A bl_gibbs handle goes back to the two passes by itself when more than a fifth of its rows leave the single pass's fast
 * path (|psi|/2 >= 1/t, n != 1: rare-event data), looked at after the 8th and 64th sweep of a chain (bl_gibbs_chain_start): a function of data and chain only.
 * bl_diag_sweep_deferred: rows the single-pass kernel handed to the full sampler since the last call (a sync). */
void bl_set_sweep_mode(int single_pass);
int  bl_diag_sweep_deferred(uint64_t *rows);
/* The coordinate sweeps of the constrained beta draw (Logit.hpp:368-399) for 64 < P <= 256 exist twice, for comparison:
 * row_split = 1 (default; env BL_BETA_SPLIT) = rows split over four wavefronts, speculative segments of 64 moves, with a
 * chain that is pressed against its bounds handed to the other kernel by itself; 0 = all rows on one wavefront, move by
 * move.  Same beta bit for bit. */
void bl_diag_beta_sweeps(int row_split);
/* The saddle-point sampler's inversion as the kernels evaluate it: out3[3i..3i+2] = (v(x_i), -log cos_rt(v), log K2(x_i))
 * from the fitted table of bl_vtab.hpp (x > 0; device pointers).  v is what the reference's v_eval returns
 * (Code/C/InvertY.cpp:57-99: table bracket + Newton solve to |dv| <= 1e-9); tests/test_inverty_ref.py compares the two. */
int bl_diag_sp_vlk_dev(double *out3, const double *x, int64_t num, void *stream);
/* ONE class pass of rpg_hybrid alone (bench.py times them one by one for the per-branch rates of config C3): cls = 4 saddle
 * point, 3 alternating series, 2 Devroye, 5 normal approximation, 1 sum of gammas; only that class's elements of x are written. */
int bl_diag_rpg_hybrid_class_dev(double *x, const double *h, const double *z, int64_t num, int cls,
                                 uint64_t seed, uint32_t epoch, uint64_t idx0, void *stream);
/* How much work the draws of a vector are (mean retry count per draw): out21[3 c + 0..2] (device, unsigned 64-bit) =
 * {observations, PG draws, Philox blocks = proposal attempts} of sampler class c (0 zero, 1 sum of gammas, 2 Devroye, 3
 * alternating series, 4 saddle point, 5 normal approximation; rpg_hybrid's dispatch, LogitWrapper.cpp:142-161; row 6 = the part of
 * row 2 with |z|/2 >= 1/t, which takes the other left-piece sampler, PolyaGamma.cpp:103).  h == NULL:
 * rpg_devroye with n = 1.  A replay of every observation's stream by the production attempt bodies, one observation per lane
 * (slow, exact: a draw is a function of its stream alone, so these are the counts of the production kernels). */
int bl_diag_count_blocks_dev(const double *h, const double *z, int64_t num, uint64_t seed, uint32_t epoch, uint64_t idx0,
                             unsigned long long *out21, void *stream);

#ifdef __cplusplus
}
#endif
#endif
