// capi_gibbs.hip -- host logic of the Gibbs / EM / mlogit drivers behind the C ABI
// (include/bayeslogit_hip.h).  Mirrors the sweep structure and slot semantics of
// Code/C/Logit.hpp:402-481 and Code/C/MultLogit.hpp:261-372; all arithmetic runs in
// the kernels of kernels_gibbs.hip.  There is no CPU compute path.
#include <cstdlib>
#include <vector>

#include "bl_gibbs_kernels.hpp"
#include "bl_host.hpp"
#include "bl_philox.hpp"

struct bl_gibbs {
  int64_t N = 0;
  int P = 0;
  uint64_t idx0 = 0, seed = 0;
  hipStream_t stream = nullptr;
  blk::SweepPlan plan;
  const double *tX = nullptr, *y = nullptr, *n = nullptr;   // not owned
  double* pool = nullptr;   // one allocation, carved below
  double *PP = nullptr, *bP = nullptr, *beta = nullptr, *beta_old = nullptr, *P0 = nullptr, *m0 = nullptr,
         *b0 = nullptr, *partial = nullptr, *colws = nullptr, *wscr = nullptr, *work = nullptr, *dist = nullptr;
  int* dead = nullptr;      // the chain's sticky "a Cholesky factorisation failed" word (in the pool; BetaArgs::dead)
  int draw_sweeps = 0, draw_sweeps_at_check = 0;   // bl_gibbs_sweep_local calls since the chain started (the single-pass fall-back policy)
};

namespace {

int num_cus()
{
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
  return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

int valid(bl_gibbs* h)
{
  if (!h) {
    blh::set_error("null gibbs handle");
    return BL_ERR_ARG;
  }
  return BL_OK;
}

}  // namespace

extern "C" {

int bl_gibbs_create(bl_gibbs** out, int64_t N_local, int P, uint64_t idx0, uint64_t seed, void* stream)
{
  if (!blh::ensure_device()) return BL_ERR_NO_DEVICE;
  if (!out || N_local < 0 || P < 1 || P > 1024) {
    blh::set_error("bl_gibbs_create: bad arguments (need N >= 0, 1 <= P <= 1024)");
    return BL_ERR_ARG;
  }
  bl_gibbs* h = new bl_gibbs;
  h->N = N_local;
  h->P = P;
  h->idx0 = idx0;
  h->seed = seed;
  h->stream = (hipStream_t)stream;
  h->plan = blk::make_plan(N_local, P, num_cus());
  const size_t PPn = (size_t)P * P;
  const size_t colws = blk::colsum_ws_doubles(N_local, P);
  const size_t wscr = (size_t)(N_local > 0 ? N_local : 1);
  const size_t work = blk::beta_work_doubles(P);
  const size_t total = 2 * PPn + 5 * (size_t)P + h->plan.partial_doubles + colws + wscr + work + 8;
  hipError_t e = hipMalloc((void**)&h->pool, total * sizeof(double));
  if (e == hipSuccess) e = hipMemsetAsync(h->pool, 0, total * sizeof(double), h->stream);
  if (e != hipSuccess) {
    blh::set_error(std::string("bl_gibbs_create: ") + hipGetErrorString(e));
    delete h;
    return BL_ERR_HIP;
  }
  double* p = h->pool;
  h->PP = p; p += PPn;
  h->P0 = p; p += PPn;
  h->bP = p; p += P;
  h->beta = p; p += P;
  h->beta_old = p; p += P;
  h->m0 = p; p += P;
  h->b0 = p; p += P;
  h->dist = p; p += 8;
  h->dead = (int*)(h->dist + 4);        // zeroed with the pool
  h->partial = p; p += h->plan.partial_doubles;
  h->colws = p; p += colws;
  h->wscr = p; p += wscr;
  h->work = p;
  *out = h;
  return BL_OK;
}

void bl_gibbs_destroy(bl_gibbs* h)
{
  if (!h) return;
  if (h->pool) (void)hipFree(h->pool);
  delete h;
}

int bl_gibbs_set_data(bl_gibbs* h, const double* tX, const double* y, const double* n)
{
  if (int rc = valid(h)) return rc;
  if (h->N > 0 && (!tX || !n)) {
    blh::set_error("bl_gibbs_set_data: null tX / n");
    return BL_ERR_ARG;
  }
  if (((uintptr_t)tX & 15) != 0) {
    blh::set_error("bl_gibbs_set_data: tX must be 16-byte aligned");
    return BL_ERR_ARG;
  }
  h->tX = tX;
  h->y = y;
  h->n = n;
  return BL_OK;
}

int bl_gibbs_set_prior(bl_gibbs* h, const double* m0, const double* P0)
{
  if (int rc = valid(h)) return rc;
  const int P = h->P;
  BL_HIP_TRY(hipMemcpyAsync(h->m0, m0, sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
  BL_HIP_TRY(hipMemcpyAsync(h->P0, P0, sizeof(double) * P * P, hipMemcpyHostToDevice, h->stream));
  BL_HIP_TRY(hipStreamSynchronize(h->stream));   // host buffers may be transient
  blk::launch_matvec(h->b0, h->P0, h->m0, P, h->stream);   // b0 = P0 m0, Logit.hpp:189
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_gibbs_set_beta(bl_gibbs* h, const double* beta)
{
  if (int rc = valid(h)) return rc;
  BL_HIP_TRY(hipMemcpyAsync(h->beta, beta, sizeof(double) * h->P, hipMemcpyHostToDevice, h->stream));
  BL_HIP_TRY(hipStreamSynchronize(h->stream));
  return BL_OK;
}

// bP_local = X_k' kappa_k, kappa_i = n_i (y_i - 1/2)   (Logit.hpp:174-183, this rank's rows)
int bl_gibbs_set_bp_local(bl_gibbs* h)
{
  if (int rc = valid(h)) return rc;
  if (!h->y && h->N > 0) {
    blh::set_error("bl_gibbs_set_bp_local: y not set");
    return BL_ERR_ARG;
  }
  blk::launch_colsum(h->tX, h->y, h->n, nullptr, nullptr, h->N, h->P, h->colws, h->bP, h->stream);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

// bP = b0 + (all-reduced) X' kappa
int bl_gibbs_finish_bp(bl_gibbs* h)
{
  if (int rc = valid(h)) return rc;
  blk::launch_vec_add(h->bP, h->bP, h->b0, h->P, h->stream);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

// A chain starts: the handle forgets what an earlier chain left behind -- the count of sweeps and of deferred rows behind
// the single-pass fall-back (so the decision is a function of THIS chain's data and draws only: the same seed on the same
// handle gives the same bits twice) and the dead-chain flag.  bl_gibbs_run / bl_gibbs_run_stream call it; a driver of the
// step API (bl_gibbs_sweep_local / bl_gibbs_draw_beta) calls it where its chain begins.
int bl_gibbs_chain_start(bl_gibbs* h)
{
  if (int rc = valid(h)) return rc;
  h->draw_sweeps = 0;
  h->draw_sweeps_at_check = 0;
  h->plan.single_pass = -1;
  BL_HIP_TRY(hipMemsetAsync(h->dead, 0, sizeof(int), h->stream));
  if ((h->P == 64 || h->P == 256) && h->N > 0)
    BL_HIP_TRY(hipMemsetAsync(blk::sweep_once_deferred_counter(h->plan, h->partial, h->N), 0, sizeof(unsigned long long),
                              h->stream));
  return BL_OK;
}

int bl_gibbs_sweep_local(bl_gibbs* h, uint32_t sweep, double* w_out)
{
  if (int rc = valid(h)) return rc;
  blk::launch_sweep(h->plan, h->tX, h->n, h->beta, nullptr, w_out, h->wscr, h->N, h->partial, h->PP, h->seed, sweep,
                    h->idx0, blk::W_DRAW, blh::status_word(h->stream), h->stream);
  BL_HIP_TRY(hipGetLastError());
  // The single-pass sweep (P = 64, 256) hands rows outside its fast path -- |psi|/2 >= 1/t, n != 1, four attempts all retries --
  // to a second kernel; when a fifth of the rows go that way the two passes are faster (rare-event data: most |psi| > 3.1).
  // Looked at after this handle's 8th and 64th sweep (one stream synchronisation each): a decision that depends on the
  // data and the chain only, so a run is reproducible; omega does not depend on it, X'Omega X in its last bits.
  h->draw_sweeps += 1;
  if ((h->P == 64 || h->P == 256) && h->N > 0 && h->plan.single_pass < 0 && blh::sweep_single_pass() &&
      (h->draw_sweeps == 8 || h->draw_sweeps == 64)) {
    unsigned long long* ctr = blk::sweep_once_deferred_counter(h->plan, h->partial, h->N);
    unsigned long long cnt = 0;
    BL_HIP_TRY(hipMemcpyAsync(&cnt, ctr, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
    BL_HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(cnt), h->stream));
    BL_HIP_TRY(hipStreamSynchronize(h->stream));
    const double per_sweep = (double)cnt / (double)(h->draw_sweeps - h->draw_sweeps_at_check);
    h->draw_sweeps_at_check = h->draw_sweeps;
    if (per_sweep > 0.2 * (double)h->N) h->plan.single_pass = 0;
  }
  return BL_OK;
}

int bl_gibbs_draw_beta(bl_gibbs* h, uint32_t sweep, int constrain)
{
  if (int rc = valid(h)) return rc;
  blk::BetaArgs a;
  a.P = h->P;
  a.PPsum = h->PP;
  a.P0 = h->P0;
  a.bP = h->bP;
  a.beta_prev = h->beta;
  a.beta_out = h->beta;
  a.work = h->work;
  a.seed = h->seed;
  a.epoch = sweep;
  a.status = blh::status_word(h->stream);
  a.dead = h->dead;
  static const bool dbg = getenv("BL_BETA_DEBUG") != nullptr;   // development aid: phase timing of the beta stage
  static unsigned long long* dbuf = nullptr;
  if (dbg) {
    if (!dbuf) (void)hipMalloc((void**)&dbuf, 32 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbuf, 0, 32 * sizeof(unsigned long long), h->stream);
    a.dbg = dbuf;
  }
  blk::launch_beta(a, constrain ? blk::B_CONSTRAINED : blk::B_MVN, h->stream);
  BL_HIP_TRY(hipGetLastError());
  if (dbg) {
    unsigned long long st[32];
    (void)hipMemcpyAsync(st, dbuf, sizeof(st), hipMemcpyDeviceToHost, h->stream);
    (void)hipStreamSynchronize(h->stream);
    // P <= 64, constrained: stamps of wavefront 0 (chol(PP) done: st[3]), of wavefront 2 (inverse done: st[4]; chol(S) from
    // st[1] to st[2]) and of the workgroup (dense stage over, sweeps start: st[7]; end: st[6])
    fprintf(stderr, "beta stage (us): chol(PP) %.1f, inverse done %.1f later, chol(S) %.1f, dense stage over at %.1f, sweeps %.1f\n",
            (st[3] - st[0]) / 100.0, st[4] ? (st[4] - st[3]) / 100.0 : 0.0, st[2] ? (st[2] - st[1]) / 100.0 : 0.0,
            st[7] ? (st[7] - st[0]) / 100.0 : 0.0, st[7] ? (st[6] - st[7]) / 100.0 : 0.0);
    if (st[7]) fprintf(stderr, "  moves redone move by move: %llu; shader clock over the sweeps: %.0f MHz\n", st[8],
                       (double)(st[10] - st[9]) / ((st[6] - st[7]) / 100.0));
    if (st[12]) fprintf(stderr, "  row-split sweeps: %llu moves with exact bounds, %llu groups (blocks through the three tests), %llu segments taken again behind an exact move\n", st[8], st[12], st[13]);
    if (st[19]) fprintf(stderr, "  shader cycles: sweeps %llu = segment set-up %llu + blocks that passed %llu + blocks with an exact move %llu + rest\n",
                        st[19], st[16], st[17], st[18]);
    if (st[19]) fprintf(stderr, "  cheap test: LDS hand-over %llu arithmetic %llu verdict exchange %llu  (segments of 64: exact moves %llu, passes taken again %llu)\n", st[20], st[21], st[22], st[20], st[21]);
    if (st[11]) fprintf(stderr, "  scan tables (wavefront 1) ready after %.1f us\n", (st[11] - st[0]) / 100.0);
    if (st[14]) fprintf(stderr, "  scans as one segment on wavefront 0: %llu shader cycles inside them\n", st[14]);
  }
  return BL_OK;
}

int bl_gibbs_em_local(bl_gibbs* h)
{
  if (int rc = valid(h)) return rc;
  blk::launch_sweep(h->plan, h->tX, h->n, h->beta, nullptr, nullptr, h->wscr, h->N, h->partial, h->PP, h->seed, 0,
                    h->idx0, blk::W_EM, blh::status_word(h->stream), h->stream);
  BL_HIP_TRY(hipGetLastError());
  return BL_OK;
}

int bl_gibbs_em_solve(bl_gibbs* h, double* dist_host)
{
  if (int rc = valid(h)) return rc;
  BL_HIP_TRY(hipMemcpyAsync(h->beta_old, h->beta, sizeof(double) * h->P, hipMemcpyDeviceToDevice, h->stream));
  blk::BetaArgs a;
  a.P = h->P;
  a.PPsum = h->PP;
  a.P0 = h->P0;
  a.bP = h->bP;
  a.beta_prev = h->beta;
  a.beta_out = h->beta;
  a.work = h->work;
  a.seed = h->seed;
  a.epoch = 0;
  a.status = blh::status_word(h->stream);
  a.dead = h->dead;
  blk::launch_beta(a, blk::B_SOLVE, h->stream);
  blk::launch_maxabsdiff(h->beta, h->beta_old, h->P, h->dist, h->stream);
  BL_HIP_TRY(hipGetLastError());
  if (dist_host) {
    BL_HIP_TRY(hipMemcpyAsync(dist_host, h->dist, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    BL_HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return BL_OK;
}

double* bl_gibbs_pp_ptr(bl_gibbs* h) { return h ? h->PP : nullptr; }
double* bl_gibbs_bp_ptr(bl_gibbs* h) { return h ? h->bP : nullptr; }
double* bl_gibbs_beta_ptr(bl_gibbs* h) { return h ? h->beta : nullptr; }

int bl_gibbs_get_beta(bl_gibbs* h, double* beta_host)
{
  if (int rc = valid(h)) return rc;
  BL_HIP_TRY(hipMemcpyAsync(beta_host, h->beta, sizeof(double) * h->P, hipMemcpyDeviceToHost, h->stream));
  BL_HIP_TRY(hipStreamSynchronize(h->stream));
  return BL_OK;
}

// Logit::gibbs, Logit.hpp:460-481: set_bP, a burn block that rewrites slot 0 `burn`
// times, then `samp` sweeps writing one slot each.  beta_out is host P x samp.
int bl_gibbs_run(bl_gibbs* h, int samp, int burn, int constrain, double* beta_out_host, double* w_out_dev)
{
  if (int rc = valid(h)) return rc;
  if (samp < 1 || burn < 0) {
    blh::set_error("bl_gibbs_run: need samp >= 1, burn >= 0");
    return BL_ERR_ARG;
  }
  const int P = h->P;
  double* hist = nullptr;
  BL_HIP_TRY(hipMalloc((void**)&hist, sizeof(double) * (size_t)P * samp));
  int rc = bl_gibbs_chain_start(h);
  if (rc == BL_OK) rc = bl_gibbs_set_bp_local(h);
  if (rc == BL_OK) rc = bl_gibbs_finish_bp(h);
  hipError_t e = hipMemsetAsync(h->beta, 0, sizeof(double) * P, h->stream);   // chain starts at beta = 0
  uint32_t sweep = 0;
  for (int m = 0; rc == BL_OK && e == hipSuccess && m < burn; ++m, ++sweep) {
    rc = bl_gibbs_sweep_local(h, sweep, w_out_dev);            // burn-in omega lands in slot 0
    if (rc == BL_OK) rc = bl_gibbs_draw_beta(h, sweep, constrain);
  }
  for (int m = 0; rc == BL_OK && e == hipSuccess && m < samp; ++m, ++sweep) {
    double* wslot = w_out_dev ? w_out_dev + (size_t)m * h->N : nullptr;
    rc = bl_gibbs_sweep_local(h, sweep, wslot);
    if (rc == BL_OK) rc = bl_gibbs_draw_beta(h, sweep, constrain);
    if (rc == BL_OK)
      e = hipMemcpyAsync(hist + (size_t)m * P, h->beta, sizeof(double) * P, hipMemcpyDeviceToDevice, h->stream);
  }
  // a failed Cholesky ends the chain (the reference throws out of draw_beta, LogitWrapper.cpp:226-229): nothing is copied out
  int st = BL_OK;
  if (rc == BL_OK && e == hipSuccess) st = blh::collect_status(h->stream);
  if (rc == BL_OK && e == hipSuccess && st != BL_ERR_NOT_PD && st != BL_ERR_HIP && beta_out_host)
    e = hipMemcpyAsync(beta_out_host, hist, sizeof(double) * (size_t)P * samp, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(hist);
  if (e != hipSuccess) {
    blh::set_error(std::string("bl_gibbs_run: ") + hipGetErrorString(e));
    return BL_ERR_HIP;
  }
  return rc != BL_OK ? rc : st;
}

// Streaming / thinning / on-device moments; same chain as bl_gibbs_run.
int bl_gibbs_run_stream(bl_gibbs* h, int samp, int burn, int constrain, int thin, double* beta_out_host, int w_mode,
                        double* w_out, const bl_gibbs_stats* stats)
{
  if (int rc = valid(h)) return rc;
  if (samp < 1 || burn < 0 || thin < 1 || w_mode < BL_W_NONE || w_mode > BL_W_ALL ||
      (w_mode != BL_W_NONE && !w_out) ||
      (stats && ((stats->w_mean_dev == nullptr) != (stats->w_var_dev == nullptr)))) {
    blh::set_error("bl_gibbs_run_stream: bad arguments");
    return BL_ERR_ARG;
  }
  const int P = h->P;
  const int64_t N = h->N;
  const int nkeep = (samp + thin - 1) / thin;
  const bool wstats = stats && stats->w_mean_dev;
  const bool bstats = stats && (stats->beta_mean_host || stats->beta_var_host);
  blh::DevBuf<double> hist, bmom, ring;
  hipError_t e = hist.alloc((size_t)P * nkeep);
  if (e == hipSuccess && bstats) e = bmom.alloc(2 * (size_t)P);
  int K = 0;
  hipStream_t cs = nullptr;
  std::vector<hipEvent_t> ev_done, ev_copied;
  if (e == hipSuccess && w_mode == BL_W_ALL) {
    const int64_t per = N > 0 ? N : 1;
    int64_t k = (int64_t)(256ll << 20) / (8 * per);                 // ring of about 256 MB, at least two slots
    K = (int)(k < 2 ? 2 : (k > samp ? samp : k));
    if (K < 1) K = 1;
    e = ring.alloc((size_t)per * K);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    ev_done.resize(K);
    ev_copied.resize(K);
    for (int i = 0; e == hipSuccess && i < K; ++i) {
      e = hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_copied[i], hipEventDisableTiming);
    }
  }
  if (e == hipSuccess && bstats) e = hipMemsetAsync(bmom.p, 0, sizeof(double) * 2 * P, h->stream);
  if (e == hipSuccess && wstats && N > 0) {
    e = hipMemsetAsync(stats->w_mean_dev, 0, sizeof(double) * N, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(stats->w_var_dev, 0, sizeof(double) * N, h->stream);
  }
  int rc = e == hipSuccess ? bl_gibbs_chain_start(h) : BL_ERR_HIP;
  if (rc == BL_OK) rc = bl_gibbs_set_bp_local(h);
  if (rc == BL_OK) rc = bl_gibbs_finish_bp(h);
  if (rc == BL_OK && hipMemsetAsync(h->beta, 0, sizeof(double) * P, h->stream) != hipSuccess) rc = BL_ERR_HIP;
  uint32_t sweep = 0;
  double* burn_slot = w_mode == BL_W_ALL ? ring.p : (w_mode == BL_W_LAST ? w_out : nullptr);
  for (int m = 0; rc == BL_OK && m < burn; ++m, ++sweep) {
    rc = bl_gibbs_sweep_local(h, sweep, burn_slot);
    if (rc == BL_OK) rc = bl_gibbs_draw_beta(h, sweep, constrain);
  }
  for (int m = 0; rc == BL_OK && e == hipSuccess && m < samp; ++m, ++sweep) {
    const int slot = K > 0 ? m % K : 0;
    double* wdst = nullptr;
    if (w_mode == BL_W_ALL) {
      wdst = ring.p + (size_t)slot * (N > 0 ? N : 1);
      if (m >= K) e = hipStreamWaitEvent(h->stream, ev_copied[slot], 0);     // the slot's previous copy has left
    } else if (w_mode == BL_W_LAST) {
      wdst = w_out;
    }
    if (e != hipSuccess) break;
    if (!wdst && wstats) wdst = h->wscr;               // the moments read omega; a sweep stores it only on request
    rc = bl_gibbs_sweep_local(h, sweep, wdst);
    if (rc != BL_OK) break;
    const double* wsrc = wdst;                                                // where this sweep's omega is
    if (wstats) blk::launch_welford(wsrc, stats->w_mean_dev, stats->w_var_dev, N, m + 1, h->stream);
    if (w_mode == BL_W_ALL && N > 0) {
      e = hipEventRecord(ev_done[slot], h->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(cs, ev_done[slot], 0);
    }
    rc = bl_gibbs_draw_beta(h, sweep, constrain);                             // enqueued before the host blocks in the copy
    if (rc != BL_OK) break;
    if (bstats) blk::launch_welford(h->beta, bmom.p, bmom.p + P, P, m + 1, h->stream);
    if (m % thin == 0 && e == hipSuccess)
      e = hipMemcpyAsync(hist.p + (size_t)(m / thin) * P, h->beta, sizeof(double) * P, hipMemcpyDeviceToDevice, h->stream);
    if (w_mode == BL_W_ALL && N > 0 && e == hipSuccess) {
      e = hipMemcpyAsync(w_out + (size_t)m * N, wdst, sizeof(double) * N, hipMemcpyDeviceToHost, cs);
      if (e == hipSuccess) e = hipEventRecord(ev_copied[slot], cs);
    }
  }
  if (rc == BL_OK && e == hipSuccess && wstats) blk::launch_welford_finish(stats->w_var_dev, N, samp, h->stream);
  if (rc == BL_OK && e == hipSuccess && bstats) blk::launch_welford_finish(bmom.p + P, P, samp, h->stream);
  // a failed Cholesky ends the chain (the reference throws out of draw_beta, LogitWrapper.cpp:226-229): no beta is copied out
  int st = BL_OK;
  if (rc == BL_OK && e == hipSuccess) st = blh::collect_status(h->stream);
  if (st == BL_ERR_NOT_PD || st == BL_ERR_HIP) rc = st;
  if (rc == BL_OK && e == hipSuccess && beta_out_host)
    e = hipMemcpyAsync(beta_out_host, hist.p, sizeof(double) * (size_t)P * nkeep, hipMemcpyDeviceToHost, h->stream);
  if (rc == BL_OK && e == hipSuccess && stats && stats->beta_mean_host)
    e = hipMemcpyAsync(stats->beta_mean_host, bmom.p, sizeof(double) * P, hipMemcpyDeviceToHost, h->stream);
  if (rc == BL_OK && e == hipSuccess && stats && stats->beta_var_host)
    e = hipMemcpyAsync(stats->beta_var_host, bmom.p + P, sizeof(double) * P, hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipStreamSynchronize(h->stream);
  if (cs) {
    hipError_t e3 = hipStreamSynchronize(cs);
    if (e2 == hipSuccess) e2 = e3;
    (void)hipStreamDestroy(cs);
  }
  for (auto& ev : ev_done) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ev_copied) if (ev) (void)hipEventDestroy(ev);
  if (e == hipSuccess) e = e2;
  if (e != hipSuccess) {
    blh::set_error(std::string("bl_gibbs_run_stream: ") + hipGetErrorString(e));
    return BL_ERR_HIP;
  }
  return rc != BL_OK ? rc : st;
}

// ================================================================ .C boundary
// gibbs(), LogitWrapper.cpp:176-234
void gibbs(double* wp, double* betap, double* yp, double* tXp, double* np, double* m0p, double* P0p, int* N, int* P,
           int* samp, int* burn)
{
  if (!blh::ensure_device()) return;
  const int64_t n = *N;
  const int p = *P;
  const uint64_t seed = bl::chain_key(blh::global_seed(), blh::next_epoch());   // bl_philox.hpp
  blh::DevBuf<double> dX, dy, dn;
  hipError_t e = dX.alloc((size_t)n * p);
  if (e == hipSuccess) e = dy.alloc(n);
  if (e == hipSuccess) e = dn.alloc(n);
  if (e == hipSuccess) e = dX.upload(tXp);
  if (e == hipSuccess) e = dy.upload(yp);
  if (e == hipSuccess) e = dn.upload(np);
  if (e != hipSuccess) {
    blh::set_error(std::string("gibbs: ") + hipGetErrorString(e));
    printf("Aborting Gibbs sampler.\n");
    return;
  }
  bl_gibbs* h = nullptr;
  int rc = bl_gibbs_create(&h, n, p, 0, seed, nullptr);
  if (rc == BL_OK) rc = bl_gibbs_set_data(h, dX.p, dy.p, dn.p);
  if (rc == BL_OK) rc = bl_gibbs_set_prior(h, m0p, P0p);
  // w (N x samp, caller-owned host memory) is filled sweep by sweep through a device ring
  if (rc == BL_OK) rc = bl_gibbs_run_stream(h, *samp, *burn, blh::global_constrain(), 1, betap, BL_W_ALL, wp, nullptr);
  if (rc != BL_OK && rc != BL_ERR_SAMPLER) {
    printf("Error: %s\n", bl_last_error());
    printf("Aborting Gibbs sampler.\n");
  }
  bl_gibbs_destroy(h);
  *N = (int)n;   // rows after merge: gibbs() itself does not merge (LogitWrapper.cpp:204)
}

// EM(), LogitWrapper.cpp:238-273 / Logit::EM, Logit.hpp:488-554
void EM(double* betap, double* yp, double* tXp, double* np, int* Np, int* Pp, double* tolp, int* max_iterp)
{
  if (!blh::ensure_device()) return;
  const int64_t n = *Np;
  const int p = *Pp;
  blh::DevBuf<double> dX, dy, dn;
  hipError_t e = dX.alloc((size_t)n * p);
  if (e == hipSuccess) e = dy.alloc(n);
  if (e == hipSuccess) e = dn.alloc(n);
  if (e == hipSuccess) e = dX.upload(tXp);
  if (e == hipSuccess) e = dy.upload(yp);
  if (e == hipSuccess) e = dn.upload(np);
  if (e != hipSuccess) {
    blh::set_error(std::string("EM: ") + hipGetErrorString(e));
    printf("Aborting EM.\n");
    return;
  }
  bl_gibbs* h = nullptr;
  int rc = bl_gibbs_create(&h, n, p, 0, 0, nullptr);   // pool is zeroed: P0 = 0, m0 = 0, beta = 0
  if (rc == BL_OK) rc = bl_gibbs_set_data(h, dX.p, dy.p, dn.p);
  if (rc == BL_OK) rc = bl_gibbs_set_bp_local(h);
  if (rc == BL_OK) rc = bl_gibbs_finish_bp(h);
  const double tol = *tolp;
  const int max_iter = *max_iterp;
  double dist = tol + 1.0;
  int iter = 0;
  while (rc == BL_OK && dist > tol && iter < max_iter) {
    rc = bl_gibbs_em_local(h);
    if (rc == BL_OK) rc = bl_gibbs_em_solve(h, &dist);
    if (rc == BL_OK) rc = blh::collect_status(nullptr);
    ++iter;
  }
  if (rc == BL_OK) {
    rc = bl_gibbs_get_beta(h, betap);
    *max_iterp = iter;
  }
  if (rc != BL_OK) {
    printf("Error: %s\n", bl_last_error());
    printf("Aborting EM.\n");
  }
  bl_gibbs_destroy(h);
}

// combine(), LogitWrapper.cpp:279-310 / Logit::compress, Logit.hpp:192-270
void combine(double* yp, double* tXp, double* np, int* N, int* P)
{
  if (!blh::ensure_device()) return;
  const int64_t n = *N;
  const int p = *P;
  if (n <= 0) return;
  blh::DevBuf<double> dX, dy, dn;
  hipError_t e = dX.alloc((size_t)n * p);
  if (e == hipSuccess) e = dy.alloc(n);
  if (e == hipSuccess) e = dn.alloc(n);
  if (e == hipSuccess) e = dX.upload(tXp);
  if (e == hipSuccess) e = dy.upload(yp);
  if (e == hipSuccess) e = dn.upload(np);
  int64_t m = n;
  int rc = BL_ERR_HIP;
  if (e == hipSuccess) rc = blk::combine_rows(dy.p, dX.p, dn.p, n, p, 1, &m, nullptr);
  if (rc == BL_OK) {
    e = hipMemcpy(yp, dy.p, sizeof(double) * m, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tXp, dX.p, sizeof(double) * m * p, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(np, dn.p, sizeof(double) * m, hipMemcpyDeviceToHost);
  }
  if (rc != BL_OK || e != hipSuccess) {
    printf("Error: %s\n", e != hipSuccess ? hipGetErrorString(e) : bl_last_error());
    printf("Aborting combine.\n");
    return;
  }
  if (m != n) {
    printf("Warning: data was combined!\n");          // Logit.hpp:248-251
    printf("N: %i, P: %i \n", (int)m, p);
  }
  *N = (int)m;
}

// mult_combine(), LogitWrapper.cpp:376-408 / MultLogit::set_data merge, MultLogit.hpp:137-208
void mult_combine(double* typ, double* tXp, double* np, int* N, int* P, int* J)
{
  if (!blh::ensure_device()) return;
  const int64_t n = *N;
  const int p = *P, u = *J - 1;
  if (n <= 0 || u < 1) return;
  blh::DevBuf<double> dX, dy, dn;
  hipError_t e = dX.alloc((size_t)n * p);
  if (e == hipSuccess) e = dy.alloc((size_t)n * u);
  if (e == hipSuccess) e = dn.alloc(n);
  if (e == hipSuccess) e = dX.upload(tXp);
  if (e == hipSuccess) e = dy.upload(typ);
  if (e == hipSuccess) e = dn.upload(np);
  int64_t m = n;
  int rc = BL_ERR_HIP;
  if (e == hipSuccess) rc = blk::combine_rows(dy.p, dX.p, dn.p, n, p, u, &m, nullptr);
  if (rc == BL_OK) {
    e = hipMemcpy(typ, dy.p, sizeof(double) * m * u, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tXp, dX.p, sizeof(double) * m * p, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(np, dn.p, sizeof(double) * m, hipMemcpyDeviceToHost);
  }
  if (rc != BL_OK || e != hipSuccess) {
    printf("Error: %s\n", e != hipSuccess ? hipGetErrorString(e) : bl_last_error());
    printf("Aborting combine.\n");
    return;
  }
  if (m != n) {
    printf("Warning: data was combined!\n");
    printf("N: %i, P: %i \n", (int)m, p);
  }
  *N = (int)m;
}

// mult_gibbs(), LogitWrapper.cpp:316-374 / MultLogit::gibbs, MultLogit.hpp:261-372.
// Data are taken as given (mlogit() merges through mlogit.combine first,
// LogitWrapper.R:371-377; the constructor's own merge is then a no-op).
void mult_gibbs(double* wp, double* betap, double* typ, double* tXp, double* np, double* m0p, double* P0p, int* N,
                int* P, int* J, int* sampp, int* burnp)
{
  if (!blh::ensure_device()) return;
  const int64_t n = *N;
  const int p = *P, u = *J - 1, samp = *sampp, burn = *burnp;
  if (u < 1 || samp < 1 || burn < 0 || p < 1) {
    printf("Error: mult_gibbs: bad arguments\nAborting Gibbs sampler.\n");
    return;
  }
  const uint64_t seed = bl::chain_key(blh::global_seed(), blh::next_epoch());   // bl_philox.hpp
  hipStream_t s = nullptr;
  const size_t PPn = (size_t)p * p;
  const size_t wslot_n = (size_t)(n > 0 ? n : 1) * u;            // doubles of omega per kept sweep
  // omega (N x (J-1) x samp in the caller's HOST memory) goes back sweep by sweep through a small device ring on a second
  // stream while the next sweep runs (as bl_gibbs_run_stream does): device memory O(N J), not O(N J samp)
  int K = (int)((int64_t)(256ll << 20) / (int64_t)(8 * wslot_n));
  K = K < 2 ? 2 : (K > samp ? samp : K);
  if (K < 1) K = 1;
  blh::DevBuf<double> dX, dty, dn, dring, dbeta, dXB, dc, deta, done, dZ, db0, dP0, dm0, db1, dxoc, dyj;
  hipError_t e = dX.alloc((size_t)n * p);
  if (e == hipSuccess) e = dty.alloc((size_t)n * u);
  if (e == hipSuccess) e = dn.alloc(n);
  if (e == hipSuccess) e = dring.alloc(wslot_n * K);
  if (e == hipSuccess) e = dbeta.alloc((size_t)p * u * samp);
  if (e == hipSuccess) e = dXB.alloc((size_t)n * (u + 1));
  if (e == hipSuccess) e = dc.alloc(n);
  if (e == hipSuccess) e = deta.alloc(n);
  if (e == hipSuccess) e = done.alloc(1);
  if (e == hipSuccess) e = dZ.alloc((size_t)p * u);
  if (e == hipSuccess) e = db0.alloc((size_t)p * u);
  if (e == hipSuccess) e = dP0.alloc(PPn * u);
  if (e == hipSuccess) e = dm0.alloc((size_t)p * u);
  if (e == hipSuccess) e = db1.alloc(p);
  if (e == hipSuccess) e = dxoc.alloc(p);
  if (e == hipSuccess) e = dyj.alloc(n);
  if (e == hipSuccess) e = dX.upload(tXp);
  if (e == hipSuccess) e = dty.upload(typ);
  if (e == hipSuccess) e = dn.upload(np);
  if (e == hipSuccess) e = dP0.upload(P0p);
  if (e == hipSuccess) e = dm0.upload(m0p);
  const double one = 1.0;
  if (e == hipSuccess) e = done.upload(&one);
  if (e == hipSuccess) e = hipMemset(dXB.p, 0, sizeof(double) * (size_t)n * (u + 1));
  if (e == hipSuccess) e = hipMemset(dbeta.p, 0, sizeof(double) * (size_t)p * u * samp);
  hipStream_t cs = nullptr;
  std::vector<hipEvent_t> ev_done(K, nullptr), ev_copied(K, nullptr);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  for (int i = 0; e == hipSuccess && i < K; ++i) {
    e = hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_copied[i], hipEventDisableTiming);
  }
  auto cleanup = [&]() {
    if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
    for (auto& ev : ev_done) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ev_copied) if (ev) (void)hipEventDestroy(ev);
  };
  if (e != hipSuccess) {
    blh::set_error(std::string("mult_gibbs: ") + hipGetErrorString(e));
    printf("Aborting Gibbs sampler.\n");
    cleanup();
    return;
  }
  bl_gibbs* h = nullptr;
  int rc = bl_gibbs_create(&h, n, p, 0, seed, s);
  if (rc == BL_OK) rc = bl_gibbs_set_data(h, dX.p, nullptr, dn.p);
  // y (U x N, category fastest) -> per-category rows for kappa: Z_j = X' (n (y_j - 1/2)), MultLogit.hpp:214-219
  for (int j = 0; rc == BL_OK && j < u; ++j) {
    blk::launch_gather_stride(dyj.p, dty.p, n, u, j, s);
    blk::launch_colsum(dX.p, dyj.p, dn.p, nullptr, nullptr, n, p, h->colws, dZ.p + (size_t)j * p, s);
    blk::launch_matvec(db0.p + (size_t)j * p, dP0.p + (size_t)j * PPn, dm0.p + (size_t)j * p, p, s);   // b0_j = P0_j m0_j
  }
  const blk::SweepPlan plan1 = blk::make_plan(n, 1, num_cus());
  const int total = burn + samp;   // burn+1 sweeps into slot 0, then samp-1 more (MultLogit.hpp:284,332)
  std::vector<char> copied_once(K, 0);
  for (int sw = 0; rc == BL_OK && e == hipSuccess && sw < total; ++sw) {
    const int slot = sw <= burn ? 0 : sw - burn;
    const int rpos = slot % K;
    double* bslot = dbeta.p + (size_t)slot * p * u;
    double* wslot = dring.p + (size_t)rpos * wslot_n;
    if (copied_once[rpos]) e = hipStreamWaitEvent(s, ev_copied[rpos], 0);      // the ring slot's previous copy has left
    if (e != hipSuccess) break;
    for (int j = 0; j < u; ++j) {
      const uint32_t epoch = (uint32_t)sw * (uint32_t)u + (uint32_t)j;
      double* bj = bslot + (size_t)j * p;
      // current beta_j lives in XB; the kernel needs beta_j itself: it is the last value written for j
      const double* bcur = (sw == 0) ? bj : (dbeta.p + (size_t)(sw <= burn ? 0 : slot - 1) * p * u + (size_t)j * p);
      // eta_j = XB_j - c_j from the stored XB, as MultLogit.hpp:293-300 has it (no pass over X); omega_j ~ PG(n, eta_j) by
      // the sweep's own psi/omega pass run on eta as an N x 1 matrix with coefficient 1 (same stream keys as any sweep)
      blk::launch_mlogit_offset(dXB.p, n, u + 1, j, dc.p, deta.p, s);
      blk::launch_sweep(plan1, deta.p, dn.p, done.p, nullptr, wslot + (size_t)j * n, h->wscr, n, nullptr, nullptr, seed,
                        epoch, 0, blk::W_DRAW, blh::status_word(s), s, 1);
      if (h->plan.fused == 1) {                   // P <= 64: X' Om X and X' Om c_j from one pass over X
        blk::launch_sweep(h->plan, dX.p, dn.p, nullptr, dc.p, wslot + (size_t)j * n, h->wscr, n, h->partial, h->PP, seed,
                          epoch, 0, blk::W_DRAW, blh::status_word(s), s, 2, dxoc.p);
      } else {
        blk::launch_sweep(h->plan, dX.p, dn.p, nullptr, nullptr, wslot + (size_t)j * n, h->wscr, n, h->partial, h->PP,
                          seed, epoch, 0, blk::W_DRAW, blh::status_word(s), s, 2);
        blk::launch_colsum(dX.p, nullptr, nullptr, wslot + (size_t)j * n, dc.p, n, p, h->colws, dxoc.p, s);   // X' Om c_j
      }
      blk::launch_vec_add(db1.p, dZ.p + (size_t)j * p, dxoc.p, p, s);
      blk::launch_vec_add(db1.p, db1.p, db0.p + (size_t)j * p, p, s);              // b1 = Z_j + X'Om c_j + b0_j
      blk::BetaArgs a;
      a.P = p;
      a.PPsum = h->PP;
      a.P0 = dP0.p + (size_t)j * PPn;
      a.bP = db1.p;
      a.beta_prev = bcur;
      a.beta_out = bj;
      a.work = h->work;
      a.seed = seed;
      a.epoch = epoch;
      a.status = blh::status_word(s);
      a.dead = h->dead;
      blk::launch_beta(a, blk::B_FROM_LIK, s);
      blk::launch_xbeta(dX.p, bj, n, p, dXB.p + (size_t)j * n, s);                 // XB_j = X beta_j
    }
    if (hipGetLastError() != hipSuccess) rc = BL_ERR_HIP;
    // the sweep's omega is final in its slot once the slot will not be rewritten: burn-in rewrites slot 0 until sweep `burn`
    if (rc == BL_OK && sw >= burn && n > 0) {
      e = hipEventRecord(ev_done[rpos], s);
      if (e == hipSuccess) e = hipStreamWaitEvent(cs, ev_done[rpos], 0);
      if (e == hipSuccess)
        e = hipMemcpyAsync(wp + (size_t)slot * wslot_n, wslot, sizeof(double) * wslot_n, hipMemcpyDeviceToHost, cs);
      if (e == hipSuccess) e = hipEventRecord(ev_copied[rpos], cs);
      copied_once[rpos] = 1;
    }
  }
  if (e != hipSuccess) {
    blh::set_error(std::string("mult_gibbs: ") + hipGetErrorString(e));
    rc = BL_ERR_HIP;
  }
  if (rc == BL_OK) rc = blh::collect_status(s);
  if (rc == BL_OK || rc == BL_ERR_SAMPLER) {
    e = dbeta.download(betap);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipStreamSynchronize(cs);
    if (e != hipSuccess) blh::set_error(std::string("mult_gibbs: ") + hipGetErrorString(e));
  } else {
    printf("Error: %s\n", bl_last_error());
    printf("Aborting Gibbs sampler.\n");
  }
  cleanup();
  bl_gibbs_destroy(h);
  *N = (int)n;
}

}  // extern "C"
