// batch_query.cpp -- host driver of the batched-query path (gemm_kernels.hip): B queries against the
// whole FLAT corpus in one GEMM pass on the matrix cores, exact top-k per query.
//
// No reference counterpart exists: VecSim answers B queries with B VecSimIndex_TopKQuery calls
// (reference src/iterators/hybrid_reader.c:374).  EVERY route re-scores the survivors of its matrix-core filter passes with the
// single-query scan's arithmetic (thresholds widened by the route's error band): replies are bit-identical to B single
// queries.  (Until round 5 the FLOAT16 / BFLOAT16 IP / cosine route -- BASELINE configs[2] itself -- handed out the MFMA sums.)
#include <atomic>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "flat_index.hpp"

namespace rsgpu {

namespace {

constexpr uint32_t kBatch = 256;  // queries per GEMM pass (the kernel's M tile)

template <typename T>
struct Dev {
  T *p = nullptr;
  size_t n = 0;
  ~Dev() {
    if (p) (void)hipFree(p);
  }
  void ensure(size_t count) {
    if (count <= n) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
};

// grow-only device scratch + pinned host mirrors of one pipeline slot of the calling thread
struct Pinned {
  void *p = nullptr;
  size_t bytes = 0;
  ~Pinned() {
    if (p) (void)hipHostFree(p);
  }
  template <typename T>
  T *ensure(size_t count) {
    if (count * sizeof(T) > bytes) {
      if (p) (void)hipHostFree(p);
      p = nullptr;
      bytes = 0;
      HIP_CHECK(hipHostMalloc(&p, count * sizeof(T), hipHostMallocDefault));
      bytes = count * sizeof(T);
    }
    return static_cast<T *>(p);
  }
};
struct BatchScratch {
  int device = -1;
  Dev<uint8_t> queries, queries16;  // (queries16: fp16 copy for the pass over an fp16 shadow; int8 copy for an int8 shadow)
  Dev<float> tau, qscale, slack_q;  // (qscale / slack_q: int8 shadow -- per-query scale product and error band)
  Dev<float> hqn;                   // (L2 pass: [2][256] |q|^2 / 2 per query, shrunk; the queries' share of the error band)
  Dev<uint32_t> cand_count, overflow, keys, out_rows, out_keys, out_n;
  Dev<uint64_t> cand, sub_cand;
  Dev<uint32_t> sub_count;
  Pinned hq, h_rows, h_keys, h_n, h_over, h_tau, h_l2;
};
// two slots: the host builds the replies of batch b while the device works on batch b+1.  Pooled, not per thread: the
// coalescer's wide passes (FlatIndex::topk_pass_wide) run on whichever caller leads, and every thread that ever led would
// otherwise keep ~150 MB of device scratch of its own.
struct BatchScratchPair {
  BatchScratch s[2];
};
std::mutex g_batch_pool_mu;
std::vector<std::unique_ptr<BatchScratchPair>> g_batch_pool;

struct BatchLease {
  std::unique_ptr<BatchScratchPair> p;
  explicit BatchLease(int device) {
    {
      std::lock_guard<std::mutex> g(g_batch_pool_mu);
      for (size_t i = 0; i < g_batch_pool.size(); i++)
        if (g_batch_pool[i]->s[0].device == device) {
          p = std::move(g_batch_pool[i]);
          g_batch_pool.erase(g_batch_pool.begin() + (long)i);
          break;
        }
    }
    if (!p) {
      p.reset(new BatchScratchPair());
      p->s[0].device = p->s[1].device = device;
    }
  }
  ~BatchLease() {
    std::lock_guard<std::mutex> g(g_batch_pool_mu);
    if (g_batch_pool.size() < 4) g_batch_pool.push_back(std::move(p));  // (more concurrent batches than that free theirs)
  }
  BatchScratch &operator[](int sl) { return p->s[sl]; }
};

}  // namespace

// which filter the last call of RSGPU_FlatIndex_TopKBatch (of any thread) ran its passes through -- a record for benches and tests
// that claim a route (RSGPU_LastBatchRoute, rsgpu_ext.h)
static std::atomic<int> g_last_batch_route{0};
int last_batch_route() { return g_last_batch_route.load(std::memory_order_relaxed); }

void release_batch_pool() {
  std::vector<std::unique_ptr<BatchScratchPair>> drop;
  {
    std::lock_guard<std::mutex> g(g_batch_pool_mu);
    drop.swap(g_batch_pool);
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (auto &p : drop) {
    (void)hipSetDevice(p->s[0].device);
    p.reset();
  }
  if (prev >= 0) (void)hipSetDevice(prev);
}

// The routes of topk_batch whose replies are bit-identical to single queries (exact re-scoring behind the matrix-core
// filter): what the coalescer may put concurrent VecSimIndex_TopKQuery calls through.  Mirrors the gating below.
bool FlatIndex::wide_pass_capable(size_t k) const {
  const ScanTuning &t = scan_tuning();
  if (!t.coalesce_wide || !t.batch_mfma || !t.gemm_qs || multi || !k || k > 1024 || key_bytes != 4) return false;
  if (__atomic_load_n(&n_rows_, __ATOMIC_RELAXED) <= (1u << 19)) return false;
  if (!batch_rescore_supported((uint32_t)(stride_ / 16))) return false;
  if (type == VecSimType_FLOAT32) {
    if (shadow_ == 1 && metric == VecSimMetric_Cosine && t.two_stage && dim <= 1024 && dim % 8 == 0 && gemm_qs_supported((uint32_t)(sstride_ / 16)))
      return true;  // fp16 shadow
    if (s8g_enabled() && metric != VecSimMetric_L2 && !h8_ && t.two_stage && gemm_qs_supported((uint32_t)(s8g_stride() / 16)) && s8g_stride() / 16 <= 64 && !s_bad_)
      return true;  // int8 rows with one scale
    if (h8_ && t.gemm_qs_f8 && metric != VecSimMetric_L2 && gemm_qs_f8_supported((uint32_t)(stride_ / 16)) && !s_bad_)
      return true;  // ... quantised in flight (round 6)
    return t.gemm_qs_f32 && gemm_qs_f32_supported((uint32_t)(stride_ / 16)) && !(metric != VecSimMetric_Cosine && hn_bad_);
  }
  // FLOAT16 / BFLOAT16: L2 through the half norms, cosine with a constant band (|x| = |q| = 1), IP with a per-query band from
  // the largest row norm -- all three re-scored exactly (round 5: IP / cosine too)
  if (type != VecSimType_FLOAT16 && type != VecSimType_BFLOAT16) return false;
  if (metric != VecSimMetric_Cosine && hn_bad_) return false;
  return gemm_qs_supported((uint32_t)(stride_ / 16));
}

void FlatIndex::topk_batch(const void *queries, size_t n_queries, size_t k, size_t *ids_out, double *scores_out,
                           size_t *counts_out, const size_t *k_each) {
  // k_each (the coalescer's wide pass): query qi wants its k_each[qi] <= k best; the passes select with k, every query takes
  // the leading k_each[qi] of its winners in (key, row) order -- the composite order makes a top-K a prefix of a top-K'
  auto k_of = [&](size_t qi) { return k_each ? std::min(k_each[qi], k) : k; };
  // FLOAT32 cosine indexes that carry an fp16 shadow (two-stage exact scan, flat_index.cpp): the MFMA filter pass
  // runs over the shadow with a 2*eps wider threshold, the survivors are re-scored from the fp32 rows with the
  // single-query scan's arithmetic -> ids and distances bit-identical to 256 single queries
  const bool via_shadow = type == VecSimType_FLOAT32 && shadow_ == 1 && metric == VecSimMetric_Cosine && !multi && k > 0 &&
                          k <= 1024 && dim <= 1024 && dim % 8 == 0 && scan_tuning().two_stage && scan_tuning().gemm_qs &&
                          gemm_qs_supported((uint32_t)(sstride_ / 16)) && batch_rescore_supported((uint32_t)(stride_ / 16));
  // int8 rows with one index-wide scale (FLOAT16 indexes: shadow_ == 3; FLOAT32 indexes created with shadow8: next to
  // their per-row shadow): the filter passes of the batch run on the int8 matrix cores
  // (round 6, h8_: FLOAT16 indexes WITHOUT the stored shadow take the same int8 passes over their fp16 rows, quantised in flight --
  // launch_gemm_qs_h8; the "two_stage" switch belongs to the stored shadows, "gemm_qs_h8" to this form)
  const bool f8 = h8_ && type == VecSimType_FLOAT32;  // (the fp32 form: knob gemm_qs_f8)
  const bool h8 = h8_ && (f8 ? scan_tuning().gemm_qs_f8 != 0 : scan_tuning().gemm_qs_h8 != 0);
  const bool s8g_shape = s8g_enabled() && metric != VecSimMetric_L2 && !multi && k > 0 && k <= 1024 && scan_tuning().gemm_qs &&
                         (h8_ ? h8 && (f8 ? gemm_qs_f8_supported((uint32_t)(stride_ / 16)) : gemm_qs_h8_supported((uint32_t)(stride_ / 16)))
                              : scan_tuning().two_stage && gemm_qs_supported((uint32_t)(s8g_stride() / 16)) &&
                                    s8g_stride() / 16 <= 64) &&  // (int8 rows up to 1024 bytes)
                         batch_rescore_supported((uint32_t)(stride_ / 16));
  // FLOAT16 / BFLOAT16 L2 indexes (round 3): the passes compute x.q on the matrix cores and fold the rows' half norms in
  // (2 (|q|^2/2 + |x|^2/2 - x.q), gemm_qs_kernels.hip "L2"); every bound is widened by the summation-order band and the
  // survivors are re-scored with the single-query L2 scan's arithmetic -> bit-identical to single queries
  const bool l2_shape = (type == VecSimType_FLOAT16 || type == VecSimType_BFLOAT16) && metric == VecSimMetric_L2 && !multi &&
                        k > 0 && k <= 1024 && scan_tuning().gemm_qs && scan_tuning().batch_mfma &&
                        gemm_qs_supported((uint32_t)(stride_ / 16)) && batch_rescore_supported((uint32_t)(stride_ / 16));
  // FLOAT32 indexes WITHOUT a shadow (round 4): the matrix-core passes read the fp32 rows themselves and round them to bf16 on
  // their way from LDS to the matrix pipe (gemm_qs_f32_kernel) -- HBM traffic is what the exact scan reads anyway, nothing is
  // stored next to the index.  Every bound is widened by the rounding band (gemm_qs_f32_rel: 2u + u^2 of |x||q| + the two
  // summation orders; cosine: |x| = |q| = 1; L2: per row through the half norms, as the 16-bit L2 passes), the survivors are
  // re-scored from the same fp32 rows with the single-query scan's arithmetic -> bit-identical to single queries
  const bool f32_shape = type == VecSimType_FLOAT32 && !via_shadow && !(s8g_shape && (h8 || scan_tuning().two_stage)) && !multi && k > 0 &&
                         k <= 1024 && scan_tuning().gemm_qs &&
                         scan_tuning().gemm_qs_f32 && scan_tuning().batch_mfma && gemm_qs_f32_supported((uint32_t)(stride_ / 16)) &&
                         batch_rescore_supported((uint32_t)(stride_ / 16));
  bool via_f32 = f32_shape && (size_t)n_rows_ + stage_n_ > (1u << 19);
  const bool l2_any = l2_shape || (f32_shape && metric == VecSimMetric_L2);
  // (an unlocked look at the size: at worst a small index computes norms it does not use, or a large one answers this batch
  // eight queries per pass -- the decision proper is taken under the lock below)
  bool via_l2 = l2_any && (size_t)n_rows_ + stage_n_ > (1u << 19) && ensure_half_norms();
  if (f32_shape && metric == VecSimMetric_L2) via_f32 = via_f32 && via_l2;
  // IP over rows that are not normalised: the band is rel |x||q| with |x| bounded by the largest row norm of the index (the
  // half norms are computed for that bound; a per-query band 2 rel |x|max |q| widens every threshold of query q)
  const bool f32_ip = f32_shape && metric == VecSimMetric_IP;
  if (f32_ip) via_f32 = via_f32 && ensure_half_norms();
  // FLOAT16 / BFLOAT16 IP / cosine WITHOUT the int8 shadow (BASELINE configs[2]; round 5): the passes' fp32 sums differ from
  // the single-query scan's by the summation order only (16-bit x 16-bit products are exact in fp32) -- every threshold is
  // widened by that band (cosine: rel x |x||q| with both norms 1; IP: per query, rel x |x|max x |q|), the survivors are
  // re-scored from the same rows with the scan's arithmetic -> bit-identical to single queries
  const bool h16_shape = (type == VecSimType_FLOAT16 || type == VecSimType_BFLOAT16) && metric != VecSimMetric_L2 && !multi && k > 0 &&
                         k <= 1024 && scan_tuning().gemm_qs && scan_tuning().batch_mfma &&
                         gemm_qs_supported((uint32_t)(stride_ / 16)) && batch_rescore_supported((uint32_t)(stride_ / 16));
  const bool h16_ip = h16_shape && metric == VecSimMetric_IP;
  // FLOAT16 IP / cosine indexes that carry the int8 shadow (shadow_ == 3): the filter passes run on the int8 matrix
  // cores over half the bytes, every threshold is widened by the query's own error band, the survivors are re-scored
  // from the fp16 rows with the single-query scan's arithmetic -> ids and distances bit-identical to single queries
  bool via_shadow8 = !via_shadow && s8g_shape && ensure_shadow8g();
  // (a 16-bit index that carries the int8 shadow takes the int8 passes; the plain 16-bit passes when the shadow is not usable)
  bool via_h16 = h16_shape && !via_shadow8 && (size_t)n_rows_ + stage_n_ > (1u << 19) && (!h16_ip || ensure_half_norms());
  const bool ip_band = f32_ip || (h16_ip && via_h16);
  const bool gemm_ok = via_shadow || via_l2 || via_f32 || via_h16 || via_shadow8;
  // a FLOAT32 index has no MFMA form of its own: without the int8 rows (too small, a non-finite row, ...) -> single queries
  const bool f32_needs_s8g = type == VecSimType_FLOAT32 && !via_shadow && !f32_shape;
  // eps of the fp16 shadow: FlatIndex::two_stage_topk; of the bf16 pass over normalised fp32 rows: |x|, |q| <= 1 + 1e-3
  const float slack = via_shadow ? 2.0f * 4e-3f
                                 : (via_f32 && metric == VecSimMetric_Cosine ? 2.0f * 1.002f * gemm_qs_f32_rel(dim)
                                                                             : (via_h16 && metric == VecSimMetric_Cosine ? 2.0f * 1.002f * hn_rel() : 0.0f));
  auto single = [&](size_t qi) {
    VecSimQueryReply *r;
    {  // (not through the coalescer: this may BE the coalescer's leader)
      flush_if_needed();
      std::shared_lock<std::shared_mutex> g(mu);
      r = topk_locked((const uint8_t *)queries + qi * elem_bytes_, k_of(qi), nullptr, BY_SCORE);
    }
    counts_out[qi] = r->len;
    for (size_t j = 0; j < r->len; j++) {
      ids_out[qi * k + j] = r->results[j].id;
      scores_out[qi * k + j] = r->results[j].score;
    }
    host_free(r->results);
    host_free(r);
  };
  // No MFMA form for this index (L2, FLOAT32 without an int8 shadow, ...): the exact multi-query scan, eight queries per
  // corpus pass (scan_mq_kernels.hip; bit-identical to single queries) -- or, failing that, one query at a time.
  auto all_single = [&]() {
    if (!mq_capable(k)) {
      for (size_t qi = 0; qi < n_queries; qi++) single(qi);
      return;
    }
    for (size_t q0 = 0; q0 < n_queries; q0 += kMqMaxQueries) {
      const size_t cnt = std::min<size_t>(kMqMaxQueries, n_queries - q0);
      TopkJob jobs[kMqMaxQueries];
      TopkJob *ptr[kMqMaxQueries];
      for (size_t i = 0; i < cnt; i++) {
        jobs[i] = TopkJob{(const uint8_t *)queries + (q0 + i) * elem_bytes_, k_of(q0 + i), nullptr, BY_SCORE};
        ptr[i] = &jobs[i];
      }
      struct Cleanup {  // (a throwing pass may leave some replies behind)
        TopkJob *j;
        size_t n;
        ~Cleanup() {
          for (size_t i = 0; i < n; i++)
            if (j[i].reply) {
              host_free(j[i].reply->results);
              host_free(j[i].reply);
            }
        }
      } cleanup{jobs, cnt};
      topk_pass(ptr, cnt);
      for (size_t i = 0; i < cnt; i++) {
        const VecSimQueryReply *r = jobs[i].reply;
        counts_out[q0 + i] = r ? r->len : 0;
        for (size_t j = 0; r && j < r->len; j++) {
          ids_out[(q0 + i) * k + j] = r->results[j].id;
          scores_out[(q0 + i) * k + j] = r->results[j].score;
        }
      }
    }
  };
  if (!gemm_ok || !scan_tuning().batch_mfma) {
    all_single();
    return;
  }
  flush_if_needed();
  std::vector<size_t> redo;
  {
    std::shared_lock<std::shared_mutex> g(mu);
    const uint32_t n = n_rows_;
    // (rows added since ensure_shadow8g, or a corpus the single-query path serves anyway: the plain fp16 passes)
    if (via_shadow8 && (s8g_built_ < n || s_bad_ || n <= (1u << 19))) via_shadow8 = false;
    if (via_l2 && (hn_built_ < n || hn_bad_ || n <= (1u << 19))) via_l2 = false;
    if (via_f32 && (n <= (1u << 19) || (metric == VecSimMetric_L2 && !via_l2) || (f32_ip && (hn_built_ < n || hn_bad_)))) via_f32 = false;
    if (via_h16 && (n <= (1u << 19) || (h16_ip && (hn_built_ < n || hn_bad_)))) via_h16 = false;
    const bool h16_type = (type == VecSimType_FLOAT16 || type == VecSimType_BFLOAT16) && metric != VecSimMetric_L2;
    if ((f32_needs_s8g && !via_shadow8) || (l2_any && !via_l2) || (f32_shape && !via_f32) || (h16_type && !via_h16 && !via_shadow8)) {
      g.unlock();
      all_single();
      return;
    }
    // the corpus the MFMA passes read
    const int g_type = via_shadow ? KT_F16 : (via_shadow8 ? KT_I8 : ktype);  // (via_f32: KT_F32 rows, launch_gemm_qs_f32)
    const size_t g_stride = via_shadow ? sstride_ : (via_shadow8 && !h8 ? s8g_stride() : stride_);
    if (!n) {
      for (size_t qi = 0; qi < n_queries; qi++) counts_out[qi] = 0;
      return;
    }
    HIP_CHECK(hipSetDevice(device));
    const size_t n_batches = (n_queries + kBatch - 1) / kBatch;
    const int n_slots = n_batches > 1 ? 2 : 1;
    CtxLease lease0(device), lease1(device);
    QueryCtx *ctxs[2] = {lease0.c, lease1.c};
    BatchLease tls_batch(device);
    const uint32_t kk = (uint32_t)std::min<size_t>(k, n);
    const uint8_t *g_rows = via_shadow ? d_shadow_ : (via_shadow8 ? s8g_rows() : d_rows_);
    const uint32_t stride16 = (uint32_t)(g_stride / 16);
    // sample prefix for the thresholds; small corpora take the all-keys path
    const bool small = n <= (1u << 19);
    uint32_t n0 = small ? n : std::min<uint32_t>(std::max<uint32_t>(round_up(n / 64, 256), 1u << 16), 1u << 18);
    n0 = std::max<uint32_t>(n0, std::min<uint32_t>(n, (uint32_t)round_up((size_t)kk * 8, 256)));
    // query-stationary filter pass (gemm_qs_kernels.hip) with PROGRESSIVE thresholds: tau comes from a small
    // sample first, the pass over the first 1/16 of the corpus re-derives it from the candidates it found
    // (the exact k-th distance of those rows), the pass over the next 3/16 does it again, and the last
    // 3/4 of the corpus is filtered with a bound ~80x tighter than the sample's: ~2.5 k candidates per query
    // instead of 6.4 k at k = 100, and the filter epilogue almost never fires.
    const bool use_qs = !small && scan_tuning().gemm_qs && kk <= 1024 &&
                        (via_f32 ? gemm_qs_f32_supported(stride16) : (via_shadow8 && h8 && f8 ? gemm_qs_f8_supported(stride16) : gemm_qs_supported(stride16)));
    if (!use_qs) {  // small corpora, K above the passes' limit: the exact multi-query scan is already cheap / the only exact form
      g.unlock();
      g_last_batch_route.store(7, std::memory_order_relaxed);
      all_single();
      return;
    }
    g_last_batch_route.store(via_f32 ? 2 : via_shadow8 ? (h8 ? (f8 ? 6 : 5) : 4) : via_shadow ? 3 : via_l2 ? 8 : via_h16 ? 1 : 0,
                             std::memory_order_relaxed);
    // FLOAT32 rows have no tiled GEMM for the sample bound: the first int8 phase runs over n0 rows with tau = +inf -- every
    // (row, query) pair becomes a candidate -- and the first bound is the K-th shadow distance among them + the band
    // (L2 passes the same way: the tiled GEMM computes 1 - x.q only)
    const bool phase0 = (via_shadow8 && type == VecSimType_FLOAT32) || via_l2 || via_f32;
    // the threshold selects drop what the new bound excludes from the lists (the L2 lists carry UPPER bounds: their test is the
    // re-scoring kernel's, on the lower bound)
    const bool prune_lists = !via_l2 && scan_tuning().batch_prune != 0;
    std::vector<uint32_t> phase_end;  // row boundaries of the filter passes
    if (use_qs) {
      n0 = std::min<uint32_t>(n, std::max<uint32_t>(1u << 15, (uint32_t)round_up((size_t)kk * 16, 256)));
      if (phase0) n0 = std::min<uint32_t>(n, std::max<uint32_t>(1u << 14, (uint32_t)round_up((size_t)kk * 16, 256)));
      // (the int8 filter's wider band makes the sample's loose bound expensive: one more, shorter first phase)
      if (n >= (1u << 23) && (scan_tuning().qs_phases == 4 || (via_shadow8 && scan_tuning().qs_phases == 0))) phase_end = {(n / 64) & ~31u, (n / 16) & ~31u, (n / 4) & ~31u, n};
      else if (n >= (1u << 23)) phase_end = {(n / 16) & ~31u, (n / 4) & ~31u, n};
      else if (n >= (1u << 21)) phase_end = {(n / 8) & ~31u, n};
      else phase_end = {n};
      if (phase0) phase_end.insert(phase_end.begin(), n0 & ~31u);
    }
    uint64_t expect_total = 4ull * kk * ((n + n0 - 1) / n0);
    if (use_qs) {
      expect_total = 0;
      uint32_t seen = n0, from = 0;
      for (uint32_t e : phase_end) {
        expect_total += (uint64_t)kk * (e - from + seen - 1) / seen + kk;
        seen = std::max(seen, e);
        from = e;
      }
      expect_total *= via_shadow || via_f32 || via_h16 ? 18 : (via_shadow8 ? 48 : 6);  // (an error band multiplies the survivors)
      if (phase0) expect_total += n0;
    }
    const uint32_t cand_cap = small ? 1 : (uint32_t)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1u << 15, expect_total));
    for (int sl = 0; sl < n_slots; sl++) {
      BatchScratch &sc = tls_batch[sl];
      sc.queries.ensure((size_t)kBatch * stride_);
      if (via_shadow || via_shadow8) sc.queries16.ensure((size_t)kBatch * (via_shadow ? sstride_ : s8g_stride()));
      if (via_f32) sc.queries16.ensure((size_t)kBatch * (stride_ / 2));
      sc.tau.ensure(kBatch);
      if (via_shadow8) {
        sc.qscale.ensure(kBatch);
        sc.slack_q.ensure(kBatch);
      }
      if (ip_band && (via_f32 || via_h16)) {
        sc.slack_q.ensure(kBatch);
        sc.h_l2.ensure<float>(2 * kBatch);
      }
      if (via_l2) {
        sc.hqn.ensure(2 * kBatch);  // (|q|^2 / 2 shrunk, then the queries' share of the band)
        sc.h_l2.ensure<float>(2 * kBatch);
      }
      sc.cand_count.ensure(kBatch);
      sc.overflow.ensure(kBatch);
      sc.keys.ensure((size_t)kBatch * n0);
      sc.out_rows.ensure((size_t)kBatch * kk);
      sc.out_keys.ensure((size_t)kBatch * kk);
      sc.out_n.ensure(kBatch);
      sc.cand.ensure((size_t)kBatch * cand_cap);
      sc.hq.ensure<uint8_t>((size_t)kBatch * stride_);
      sc.h_rows.ensure<uint32_t>((size_t)kBatch * kk);
      sc.h_keys.ensure<uint32_t>((size_t)kBatch * kk);
      sc.h_n.ensure<uint32_t>(kBatch);
      sc.h_over.ensure<uint32_t>(kBatch);
    }
    // per-(workgroup, query, lane half) sub-lists of a pass, 8x the expected length of the fullest pass
    uint32_t sub_cap = 32, qs_grid_max = 0;
    if (use_qs) {
      uint32_t seen = n0, from = 0;
      for (uint32_t e : phase_end) {
        const uint32_t grid = gemm_qs_grid(e - from);
        qs_grid_max = std::max(qs_grid_max, grid);
        const uint64_t expect = (uint64_t)kk * ((e - from + seen - 1) / seen) / (2ull * grid) + 1;
        while (sub_cap < (via_shadow || via_f32 || via_h16 ? 24 : (via_shadow8 ? 64 : 8)) * expect) sub_cap *= 2;
        if (phase0 && from == 0)  // every row of the first phase lands in a sub-list: 16 per lane and tile
          while (sub_cap < 16u * (((e + 31) / 32 + grid - 1) / grid)) sub_cap *= 2;
        seen = std::max(seen, e);
        from = e;
      }
      for (int sl = 0; sl < n_slots; sl++) {
        tls_batch[sl].sub_count.ensure((size_t)qs_grid_max * kBatch * 2);
        tls_batch[sl].sub_cand.ensure((size_t)qs_grid_max * kBatch * 2 * sub_cap);
      }
    }
    const bool prof = scan_profile().enabled.load(std::memory_order_relaxed) != 0;
    std::vector<uint8_t> l2_redo[2];  // per slot: queries of the batch the L2 pass leaves to the exact scan

    // everything of batch q0.. onto the slot's stream, nothing waited for
    // (RSGPU_DEBUG_SYNC=1: wait after every step and name it on stderr -- which kernel of the pipeline faulted)
    static const bool dbg_sync = getenv("RSGPU_DEBUG_SYNC") != nullptr;
    auto enqueue = [&](int sl, size_t q0) {
      BatchScratch &sc = tls_batch[sl];
      // ONE stream carries every batch of the call (round 6; the slots own scratch, pinned mirrors and events only): two batches on
      // two streams interleaved their kernels -- 3.49 ms per pass where a pass alone takes 3.16 -- and the host is ahead anyway
      QueryCtx *c = ctxs[0];
      auto dbg = [&](const char *what) {
        if (!dbg_sync) return;
        fprintf(stderr, "[batch] %s ...", what);
        const hipError_t e = hipStreamSynchronize(c->stream);
        fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
      };
      const uint32_t nb = (uint32_t)std::min<size_t>(kBatch, n_queries - q0);
      uint8_t *hq = static_cast<uint8_t *>(sc.hq.p);
      memset(hq, 0, (size_t)kBatch * stride_);  // unused query rows stay zero (their results are ignored)
      for (uint32_t i = 0; i < nb; i++) {
        uint8_t *dst = hq + (size_t)i * stride_;
        memcpy(dst, (const uint8_t *)queries + (q0 + i) * elem_bytes_, elem_bytes_);
        if (metric == VecSimMetric_Cosine) normalize_host(dst);
      }
      HIP_CHECK(hipMemcpyAsync(sc.queries.p, hq, (size_t)kBatch * stride_, hipMemcpyHostToDevice, c->stream));
      const uint8_t *g_queries = sc.queries.p;
      if (via_shadow) {  // fp16 (RNE) copy of the normalised queries, made by the kernel that makes the shadow rows
        launch_shadow_rows(sc.queries.p, stride_, (uint32_t)dim, 0, kBatch, sc.queries16.p, sstride_, c->stream);
        g_queries = sc.queries16.p;
      }
      if (via_f32) {  // bf16 copies of the (normalised) queries, rounded as the pass rounds the rows
        launch_convert_queries_bf16(sc.queries.p, stride_, (uint32_t)dim, kBatch, sc.queries16.p, stride_ / 2, c->stream);
        g_queries = sc.queries16.p;
      }
      const float *slack_q = nullptr, *qscale = nullptr;
      if (via_shadow8) {  // int8 copies of the queries with their own scales + the per-query error band
        launch_quantize_queries(ktype, sc.queries.p, stride_, (uint32_t)dim, kBatch, s8g_scale_, d_s8g_stats_, sc.queries16.p,
                                s8g_stride(), sc.qscale.p, sc.slack_q.p, c->stream);
        dbg("quantize_queries");
        g_queries = sc.queries16.p;
        slack_q = sc.slack_q.p;
        qscale = sc.qscale.p;
      }
      if (ip_band && (via_f32 || via_h16)) {  // per-query band: 2 rel |x|max |q|; a query whose norm is not finite goes to the exact scan
        float *hl = static_cast<float *>(sc.h_l2.p);
        const float rel = (via_f32 ? gemm_qs_f32_rel(dim) : hn_rel()) * 1.002f, xmax = sqrtf(2.0f * hn_max_) * 1.000001f;
        l2_redo[sl].assign(kBatch, 0);
        for (uint32_t i = 0; i < kBatch; i++) {
          const float h = i < nb ? half_sq_norm_host((const uint8_t *)queries + (q0 + i) * elem_bytes_) : 0.0f;
          if (!(h <= 3.0e38f)) l2_redo[sl][i] = 1;
          hl[i] = h <= 3.0e38f ? 2.0f * rel * xmax * sqrtf(2.0f * h) * 1.000001f : 0.0f;
        }
        HIP_CHECK(hipMemcpyAsync(sc.slack_q.p, hl, kBatch * sizeof(float), hipMemcpyHostToDevice, c->stream));
        slack_q = sc.slack_q.p;
      }
      const float *l2_hn = nullptr, *l2_hq = nullptr;
      RowBand rb;
      if (via_l2) {
        // |d~ - d| <= rel (hn[row] + hq): the MFMA's summation order against the scan's, both within dim roundings of |x||q|
        // <= hn + hq, the rows' norms within dim roundings of hn (docs/DESIGN_NOTES.md section 3 "L2 on the matrix cores").  The pass
        // works with norms and hq shrunk by (1 - rel/2): what it emits is the lower bound d~ - band(row, q); the candidate
        // lists carry the upper bound lb + 2 band(row, q), the thresholds are selected from those without further slack.
        float *hl = static_cast<float *>(sc.h_l2.p);
        const float rel = hn_rel(), shrink = 1.0f - 0.5f * rel;
        l2_redo[sl].assign(kBatch, 0);
        for (uint32_t i = 0; i < kBatch; i++) {
          float h = i < nb ? half_sq_norm_host((const uint8_t *)queries + (q0 + i) * elem_bytes_) : 0.0f;
          if (!(h <= 3.0e38f)) {  // a query with an inf / NaN element: answered by the exact scan (finalize)
            l2_redo[sl][i] = 1;
            h = 0.0f;
          }
          hl[i] = h * shrink;
          hl[kBatch + i] = 2.0f * rel * h;
        }
        HIP_CHECK(hipMemcpyAsync(sc.hqn.p, hl, 2 * kBatch * sizeof(float), hipMemcpyHostToDevice, c->stream));
        l2_hn = d_hnorm_;
        l2_hq = sc.hqn.p;
        rb.hnorm = d_hnorm_;
        rb.hq2 = sc.hqn.p + kBatch;
        rb.c1 = 2.0f * rel / shrink;
        rb.inv2rel = 0.5f / rel;
      }
      if (prof) HIP_CHECK(hipEventRecord(ctxs[sl]->ev0, c->stream));
      {
        // the sample's bound: from the exact rows when the filter runs on the int8 shadow (the tiled GEMM has no int8 form;
        // an exact bound widened by the band is as good as a shadow bound widened by it)
        if (phase0) {  // tau = +inf for the queries of the batch, -inf for the padding
          float *ht = sc.h_tau.ensure<float>(kBatch);  // (pinned, one per slot: free again once the slot's batch is finalized)
          for (uint32_t i = 0; i < kBatch; i++)
            ht[i] = i < nb && !((via_l2 || ip_band) && l2_redo[sl][i]) ? __builtin_inff() : -__builtin_inff();
          HIP_CHECK(hipMemcpyAsync(sc.tau.p, ht, kBatch * sizeof(float), hipMemcpyHostToDevice, c->stream));
        } else {
          if (via_shadow8)
            launch_gemm_topk(ktype, d_rows_, sc.queries.p, (uint32_t)(stride_ / 16), 0, n0, 0, sc.keys.p, n0, nullptr, nullptr,
                             nullptr, 0, c->stream);
          else
            launch_gemm_topk(g_type, g_rows, g_queries, stride16, 0, n0, 0, sc.keys.p, n0, nullptr, nullptr, nullptr, 0, c->stream);
          launch_batch_threshold(sc.keys.p, n0, n0, kk, kBatch, nb, sc.tau.p, c->stream, 1, slack, slack_q);
        }
        HIP_CHECK(hipMemsetAsync(sc.cand_count.p, 0, kBatch * sizeof(uint32_t), c->stream));
        HIP_CHECK(hipMemsetAsync(sc.overflow.p, 0, kBatch * sizeof(uint32_t), c->stream));
        {
          uint32_t from = 0;
          for (size_t ph = 0; ph < phase_end.size(); ph++) {
            const uint32_t e = phase_end[ph];
            if (!(via_f32 ? launch_gemm_qs_f32(g_rows, g_queries, stride16, from, e, sc.tau.p, sc.sub_count.p, sc.sub_cand.p, sub_cap,
                                               c->stream, l2_hn, l2_hq)
                  : via_shadow8 && h8
                      ? (f8 ? launch_gemm_qs_f8(g_rows, g_queries, stride16, from, e, sc.tau.p, sc.sub_count.p, sc.sub_cand.p, sub_cap, c->stream,
                                                qscale, f8_inv_)
                            : launch_gemm_qs_h8(g_rows, g_queries, stride16, from, e, sc.tau.p, sc.sub_count.p, sc.sub_cand.p, sub_cap, c->stream,
                                                qscale, h8_inv_bits_))
                          : launch_gemm_qs(g_type, g_rows, g_queries, stride16, from, e, sc.tau.p, sc.sub_count.p, sc.sub_cand.p, sub_cap,
                                           c->stream, qscale, l2_hn, l2_hq)))
              throw std::runtime_error("batched pass: the matrix-core kernel refused a row shape the route was gated on");
            dbg("filter pass");
            launch_compact_cand(sc.sub_count.p, sc.sub_cand.p, sub_cap, gemm_qs_grid(e - from), sc.cand_count.p,
                                sc.cand.p, cand_cap, ph > 0, c->stream, via_l2 ? &rb : nullptr);
            dbg("compact");
            if (ph + 1 < phase_end.size())
              launch_batch_threshold_cand(sc.cand.p, sc.cand_count.p, cand_cap, kk, kBatch, nb, sc.tau.p, sc.overflow.p,
                                          c->stream, slack, slack_q, prune_lists);
            dbg("threshold");
            from = e;
          }
        }
        {
          // final band: tau = exact k-th shadow distance of the whole corpus + 2 eps; the candidates inside it get
          // their exact keys (the single-query scan's arithmetic), then the usual exact select over (key, row)
          launch_batch_threshold_cand(sc.cand.p, sc.cand_count.p, cand_cap, kk, kBatch, nb, sc.tau.p, sc.overflow.p,
                                      c->stream, slack, slack_q, prune_lists);
          if (!launch_batch_rescore(d_rows_, stride_, n, sc.queries.p, stride_, sc.cand.p, sc.cand_count.p, cand_cap, kBatch,
                                    sc.tau.p, c->stream, via_shadow8 || via_l2 || via_h16 ? ktype : KT_F32, via_l2 ? KM_L2 : KM_IP,
                                    via_l2 ? &rb : nullptr, prune_lists))
            throw std::runtime_error("batched shadow pass: the re-scoring kernel refused a row shape the route was gated on");
          dbg("rescore");
        }
        launch_batch_select_cand(sc.cand.p, sc.cand_count.p, cand_cap, kk, kBatch, sc.out_rows.p, sc.out_keys.p,
                                 sc.out_n.p, kk, sc.overflow.p, c->stream);
      }
      HIP_CHECK(hipGetLastError());
      if (prof) HIP_CHECK(hipEventRecord(ctxs[sl]->ev1, c->stream));
      HIP_CHECK(hipMemcpyAsync(sc.h_rows.p, sc.out_rows.p, (size_t)kBatch * kk * 4, hipMemcpyDeviceToHost, c->stream));
      HIP_CHECK(hipMemcpyAsync(sc.h_keys.p, sc.out_keys.p, (size_t)kBatch * kk * 4, hipMemcpyDeviceToHost, c->stream));
      HIP_CHECK(hipMemcpyAsync(sc.h_n.p, sc.out_n.p, kBatch * 4, hipMemcpyDeviceToHost, c->stream));
      HIP_CHECK(hipMemcpyAsync(sc.h_over.p, sc.overflow.p, kBatch * 4, hipMemcpyDeviceToHost, c->stream));
      if (!prof) HIP_CHECK(hipEventRecord(ctxs[sl]->ev1, c->stream));  // the slot's answers are in pinned memory behind this one
    };

    // (RSGPU_BATCH_TRACE=1: host timestamps of the pipeline on stderr -- where a call's wall time goes)
    static const bool trace = getenv("RSGPU_BATCH_TRACE") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what, size_t b) {
      if (trace) fprintf(stderr, "[batch-trace] %-14s %zu  %8.3f ms\n", what, b,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    };
    // wait for the slot's batch and build its replies
    auto finalize = [&](int sl, size_t q0) {
      BatchScratch &sc = tls_batch[sl];
      QueryCtx *c = ctxs[sl];
      const uint32_t nb = (uint32_t)std::min<size_t>(kBatch, n_queries - q0);
      const uint32_t *h_rows = static_cast<const uint32_t *>(sc.h_rows.p), *h_keys = static_cast<const uint32_t *>(sc.h_keys.p);
      const uint32_t *h_n = static_cast<const uint32_t *>(sc.h_n.p), *h_over = static_cast<const uint32_t *>(sc.h_over.p);
      if (prof) HIP_CHECK(hipStreamSynchronize(ctxs[0]->stream));  // (ev1 sits in front of the copies there)
      else HIP_CHECK(hipEventSynchronize(c->ev1));
      stamp("synced", q0 / kBatch);
      if (prof) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) {
          ScanProfile &pf = scan_profile();
          pf.launches++;
          pf.bytes += (uint64_t)n * (via_shadow ? dim * 2 : (via_shadow8 && !h8 ? dim : elem_bytes_));
          pf.nanos += (uint64_t)((double)ms * 1e6);
        }
      }
      std::vector<VecSimQueryResult> res;
      // the winners' labels are random reads of a table of 8 B per row (80 MB at 10 M rows): requested a query ahead, 100 misses in
      // flight instead of one after the other (1.25 ms per 256 x 100 replies before: a third of a pass)
      auto prefetch_labels = [&](uint32_t i) {
        const uint32_t m = std::min(h_n[i], kk);
        for (uint32_t j = 0; j < m; j++) {
          const uint32_t r = h_rows[(size_t)i * kk + j];
          if (r < row_label_.size()) __builtin_prefetch(&row_label_[r]);
        }
      };
      if (nb) prefetch_labels(0);
      for (uint32_t i = 0; i < nb; i++) {
        const size_t qi = q0 + i;
        if (i + 1 < nb) prefetch_labels(i + 1);
        if (h_over[i] || ((via_l2 || ip_band) && l2_redo[sl][i])) {  // candidate list overflowed (or a non-finite L2 query): redo this query on the single-query path
          redo.push_back(qi);
          continue;
        }
        const uint32_t got = std::min(h_n[i], kk);
        if (got < kk) {  // fewer winners than rows asked for: NaN distances the filter passes drop -- the exact scan ranks them last
          redo.push_back(qi);
          continue;
        }
        // the device hands back the exact top-k SET (selected by (key, row)); reply order: (score, label) ascending
        uint32_t take = got;
        if (k_each && k_of(qi) < got) {  // the leading k_each of the winners in the selection's own order
          take = (uint32_t)k_of(qi);
          std::vector<std::pair<uint32_t, uint32_t>> kr(got);
          for (uint32_t j = 0; j < got; j++) kr[j] = {h_keys[(size_t)i * kk + j], h_rows[(size_t)i * kk + j]};
          std::sort(kr.begin(), kr.end());
          res.resize(take);
          for (uint32_t j = 0; j < take; j++) res[j] = VecSimQueryResult{(size_t)label_at(kr[j].second), score_of(kr[j].first)};
        } else {
          res.resize(got);
          for (uint32_t j = 0; j < got; j++)
            res[j] = VecSimQueryResult{(size_t)label_at(h_rows[(size_t)i * kk + j]), score_of(h_keys[(size_t)i * kk + j])};
        }
        const auto before = [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return score_id_before(a.score, a.id, b.score, b.id); };
        if (!std::is_sorted(res.begin(), res.end(), before)) std::sort(res.begin(), res.end(), before);
        counts_out[qi] = take;
        for (uint32_t j = 0; j < take; j++) {
          ids_out[qi * k + j] = res[j].id;
          scores_out[qi * k + j] = res[j].score;
        }
      }
    };

    for (size_t b = 0; b < n_batches; b++) {
      stamp("enqueue", b);
      enqueue((int)(b & 1) % n_slots, b * kBatch);
      stamp("enqueued", b);
      if (b > 0) finalize((int)((b - 1) & 1), (b - 1) * kBatch);
      if (b > 0) stamp("finalized", b - 1);
    }
    finalize((int)((n_batches - 1) & 1) % n_slots, (n_batches - 1) * kBatch);
    stamp("finalized", n_batches - 1);
  }
  for (size_t qi : redo) single(qi);
}

}  // namespace rsgpu
