// flat_index.hpp -- host side of the MI355X FLAT index: HBM-resident corpus, label maps, per-query
// workspaces and the query drivers that string the HIP kernels together.
//
// What lives where (DESIGN.md section 2 "data layout"):
//   HBM   rows      [cap_rows][stride]  row-contiguous, stride = dim*sizeof(T) rounded up to 16 B,
//                                       zero padded; cosine rows are stored normalised
//         labels    [cap_rows] u64      row -> label (doc id)
//         per query keys [n] u32 (orderable distance), 8x256 histograms, <=K winners
//   host  row_label vector, label -> row(s) hash map, a pinned staging block for AddVector
// Rows are dense: DeleteVector moves the last row into the hole ([upstream-memory D8]).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "VecSim/vec_sim.h"
#include "common.hpp"
#include "grow_buffer.hpp"
#include "kernels.hpp"
#include "label_table.hpp"

namespace rsgpu {

struct Hit {
  uint32_t row;
  uint64_t key;
};
// exclusive lower bound / inclusive upper bound of a selection in composite (key,row) order
struct Bound {
  uint64_t key = 0;
  uint32_t row = 0;
  bool valid = false;
};

// Per in-flight query workspace: own stream, device scratch and pinned host mirrors.
struct QueryCtx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // query blob
  uint8_t *d_query = nullptr, *h_query = nullptr;
  size_t query_cap = 0;
  uint64_t cached_query_owner = 0;  // index uid whose query is resident in d_query (adhoc cache)
  size_t cached_query_len = 0;
  // keys
  uint32_t *d_keys = nullptr;
  size_t keys_cap = 0;
  // select state
  uint32_t *d_hist = nullptr, *d_counters = nullptr;
  uint64_t *d_bound = nullptr;
  uint32_t *d_out_rows = nullptr, *h_out_rows = nullptr;
  uint64_t *d_out_keys = nullptr, *h_out_keys = nullptr;  // sized for u64 keys, u32 keys use the front half
  size_t out_cap = 0;
  uint32_t *h_counters = nullptr;  // [4] counters + bound (2 x u64) behind them
  // gather
  uint32_t *d_ids = nullptr, *h_ids = nullptr;
  float *d_dists = nullptr, *h_dists = nullptr;
  size_t gather_cap = 0;
  // threshold-filter select (small K): tau, candidate list (row,key), its counter / overflow flag /
  // winner count, and their pinned mirror [0]=out_n [1]=overflow [2]=cand_count
  static constexpr uint32_t kCandCap = 1u << 16;
  float *d_tau = nullptr;
  uint64_t *d_cand = nullptr;
  uint32_t *d_fcnt = nullptr;  // [0] cand_count [1] overflow [2] out_n
  uint32_t *h_fcnt = nullptr;
  // coalesced pass (several queries per scan, scan_mq_kernels.hip): the queries [B][stride], per-query threshold /
  // candidate list / counter, and the pinned per-query winner count + overflow flag
  uint8_t *d_mq_queries = nullptr, *h_mq_queries = nullptr;
  size_t mq_query_cap = 0;
  float *d_mq_tau = nullptr;      // [kMqMaxQueries]
  uint32_t *d_mq_cnt = nullptr;   // [kMqMaxQueries]
  uint64_t *d_mq_cand = nullptr;  // [kMqMaxQueries][kCandCap]
  uint32_t *h_mq_n = nullptr, *h_mq_over = nullptr;  // [kMqMaxQueries] each (one pinned block)
  void ensure_mq(size_t query_bytes);
  // scan profiling (events read back after the query's own sync)
  bool prof_pending = false;
  uint32_t prof_rows = 0;
  size_t prof_bytes_per_row = 0;

  explicit QueryCtx(int dev);
  ~QueryCtx();
  void ensure_query(size_t bytes);
  void ensure_keys(size_t words);  // u32 words: rows x key_bytes/4
  void ensure_out(size_t k);
  void ensure_gather(size_t m);
  uint64_t *h_bound() { return reinterpret_cast<uint64_t *>(h_counters + 4); }
};

class CtxPool {
 public:
  static CtxPool &get();
  QueryCtx *acquire(int device);
  void release(QueryCtx *c);
  void drain();  // frees every idle workspace (tests / VecSim_GetSharedMemory accounting)
  size_t bytes() const { return bytes_.load(); }
  void account(long delta) { bytes_ += delta; }

 private:
  std::mutex mu_;
  std::vector<QueryCtx *> idle_;
  std::atomic<long> bytes_{0};
};

struct CtxLease {
  QueryCtx *c;
  explicit CtxLease(int device) : c(CtxPool::get().acquire(device)) {}
  ~CtxLease() {
    if (c) CtxPool::get().release(c);
  }
  QueryCtx *operator->() { return c; }
  QueryCtx *release() {
    QueryCtx *r = c;
    c = nullptr;
    return r;
  }
};

// Scan-kernel profile, filled when profiling is on (bench.py's roofline leg).
struct ScanProfile {
  std::atomic<uint64_t> launches{0};
  std::atomic<uint64_t> bytes{0};
  std::atomic<uint64_t> nanos{0};
  std::atomic<int> enabled{0};
};
ScanProfile &scan_profile();

// What became of the two-stage (shadow) scans of this process: a query that leaves the two-stage path runs the plain
// fp32 scan -- exact, but four times the bytes -- so every way out is counted (RSGPU_GetTwoStageStats; bench.py puts the
// counters into its two_stage sub-record).
struct TwoStageStats {
  enum { ATTEMPTS, OK, FB_SHAPE, FB_QUERY_OR_BAND, FB_FIRST_PASS_OVERFLOW, FB_BAND_OVERFLOW, FB_TOO_FEW, FB_SELECT, N };
  std::atomic<uint64_t> v[N];
  TwoStageStats() { for (auto &x : v) x = 0; }
};
TwoStageStats &two_stage_stats();

// One VecSimIndex_TopKQuery call on its way through the coalescer (FlatIndex::topk): the caller's arguments, and what the
// pass that served it left behind.
struct TopkJob {
  const void *query;
  size_t k;
  void *tctx;
  VecSimQueryReply_Order order;
  VecSimQueryReply *reply = nullptr;
  std::exception_ptr err;
  bool done = false;
  // the coalescer's jobs (FlatIndex::topk): the timeout callback is the CALLER's to poll -- on its own thread, as the
  // single-query path does (a host callback may rely on thread-local state), every millisecond while it waits in the queue --
  // so the pass that serves the job does not poll it; `taken`: a leader has put the job into its pass (the caller can no
  // longer leave the queue; it checks its timeout again when the pass is done)
  bool owner_polls = false, taken = false;
};

// What the coalescer did (RSGPU_GetCoalesceStats; bench.py's concurrent_callers sub-record).
struct CoalesceStats {
  std::atomic<uint64_t> passes{0}, queries{0}, mq_passes{0}, mq_queries{0}, lingers{0}, linger_ns{0}, mq_device_ns{0},
      mq_redo{0}, wide_passes{0}, wide_queries{0}, left_queue{0};
};
CoalesceStats &coalesce_stats();

class FlatIndex {
 public:
  explicit FlatIndex(const BFParams &p, void *log_ctx);
  ~FlatIndex();

  // writes (single writer, excluded from readers by the caller; guarded again here)
  int add(const void *blob, size_t label);
  int remove(size_t label);
  void reserve(size_t rows);
  int add_device_rows(const void *dev_rows, size_t n, size_t first_label);
  // n synthetic rows generated in place (corpus_kernels.hip): row i = Philox(seed; first_index + i)
  int add_philox_rows(uint64_t seed, uint64_t first_index, size_t n, size_t first_label);
  // stored rows [row_begin, row_begin+n) as they are in HBM (cosine: normalised), tightly packed, to host memory
  void read_rows(uint32_t row_begin, size_t n, void *host_out);

  size_t size();
  size_t label_count();
  bool contains(size_t label);  // a vector is stored (committed or staged) under this label
  VecSimIndexBasicInfo basic_info() const;
  size_t memory() const;

  // queries
  VecSimQueryReply *topk(const void *query, size_t k, VecSimQueryParams *qp, VecSimQueryReply_Order order);
  VecSimQueryReply *range(const void *query, double radius, VecSimQueryParams *qp, VecSimQueryReply_Order order);
  double distance_from(size_t label, const void *normalized_blob);
  // B queries at once (non-ABI): ids_out/scores_out are [B][k], counts_out[B] the hits per query.
  // fp16/bf16 IP/cosine run as a GEMM on the matrix cores (batch_query.cpp); everything else loops
  // over topk().
  void topk_batch(const void *queries, size_t n_queries, size_t k, size_t *ids_out, double *scores_out,
                  size_t *counts_out, const size_t *k_each = nullptr);  // k_each[qi] <= k: per-query K (the coalescer's wide pass)
  // topk_batch would answer a K = k query of this index through a matrix-core filter pass + exact re-scoring (replies
  // bit-identical to single queries): the coalescer may then put up to kWidePass concurrent calls into one pass
  bool wide_pass_capable(size_t k) const;
  static constexpr uint32_t kWidePass = 256;
  size_t wide_min() const;  // passes of at least this many capable calls take the matrix-core form (knob coalesce_wide_min)
  // up to kMqMaxQueries queries in ONE pass over the corpus (the coalescer's pass; also what topk_batch uses for indexes
  // without an MFMA form): fills job->reply (or job->err) of every job.  Replies are bit-identical to topk()'s.
  void topk_pass(TopkJob *const *jobs, size_t n_jobs);
  bool mq_capable(size_t k) const;   // the multi-query scan can serve a top-k query of this index
  bool coalescible(size_t k) const;  // ... and concurrent calls are worth coalescing (knobs "coalesce" / "coalesce_min_mib", corpus size)
  // how long a pass may wait for the callers of the previous pass to come back (knob, or 5 % of a pass: 20..300 us)
  int coalesce_linger_us() const;
  bool prefer_adhoc(size_t subset, size_t k, bool initial_check);
  // the decision itself: N vectors under `labels` labels, `subset` of them pass the filter
  static bool prefer_adhoc_rule(size_t N, size_t labels, size_t dim, size_t subset);

  // building blocks shared with the batch iterator / adhoc ctx / device-output extension
  void flush();                                                 // staged adds -> HBM (unique lock held)
  void flush_if_needed();                                       // takes the locks itself
  void upload_query(QueryCtx *c, const void *blob, bool normalize);
  void scan_all(QueryCtx *c, uint32_t n);                        // keys for rows [0,n)
  // exact selection of the k smallest (key,row) composites of the scan's keys above `lower`
  void select(QueryCtx *c, uint32_t n, uint32_t k, const Bound &lower, std::vector<Hit> &out, Bound *upper);
  void gather(QueryCtx *c, const size_t *labels, size_t m, double *out);
  // label of a row the DEVICE named (a selected / filtered row index read back from a kernel): a row outside the index
  // would be a kernel bug -- reported, never read past the table
  uint64_t label_at(uint32_t row) const {
    if (row >= row_label_.size())
      throw std::runtime_error("the device returned row " + std::to_string(row) + " of an index of " + std::to_string(row_label_.size()));
    return row_label_[row];
  }
  size_t label_of_row(uint32_t row) const { return (size_t)label_at(row); }
  // labels are base + row for every row (no table at all)
  bool identity_labels(uint64_t *base) const { return labels_.identity(base); }
  // label -> row as the kernels take it (label_table.hpp); false: the labels are too sparse for a device table and the
  // caller translates on the host (first_row_of / gather)
  bool device_label_rows(LabelRows *out) const { return labels_.device_view(n_rows_, out); }
  int label_mode() const { return (int)labels_.mode(); }
  // first committed row of a label, 0xFFFFFFFF when absent
  uint32_t first_row_of(size_t label) const {
    std::vector<uint32_t> r;
    rows_of(label, r);
    for (uint32_t x : r)
      if (x < n_rows_) return x;
    return 0xFFFFFFFFu;
  }
  uint32_t committed_rows() const { return n_rows_; }
  const uint64_t *device_labels() const { return d_labels_; }
  const void *device_rows() const { return d_rows_; }
  size_t stride() const { return stride_; }

  VecSimType type;
  VecSimMetric metric;
  size_t dim;
  bool multi;
  size_t block_size;
  int ktype, kmetric;
  int key_bytes;  // 4: fp32 distances / u32 keys, 8: fp64 distances / u64 keys (FLOAT64)
  double score_of(uint64_t key) const { return key_bytes == 8 ? key_to_dist64(key) : (double)key_to_dist((uint32_t)key); }
  int device;
  uint64_t uid;
  void *log_ctx;
  std::atomic<int> last_mode{EMPTY_MODE};
  // bumped whenever a committed row changes place (DeleteVector moves the last row into the hole): per-row state
  // computed before the bump (a batch iterator's keys) no longer lines up with row -> label
  std::atomic<uint64_t> layout_epoch{0};
  std::shared_mutex mu;

 private:
  void grow(size_t min_rows);
  void normalize_host(void *blob) const;
  void check_bulk_labels(size_t n, size_t first_label) const;
  void commit_bulk_rows(size_t n, size_t first_label);  // normalise/shadow/label the n rows written behind n_rows_
  void rows_of(size_t label, std::vector<uint32_t> &out) const { labels_.rows_of(label, out); }

  size_t elem_bytes_, stride_;
  // optional low-precision shadow of the rows (FLOAT32, single-value; chosen at creation by ScanTuning::shadow16 --
  // cosine only -- / shadow8 -- cosine, IP, L2): 0 none, 1 fp16, 2 int8 with {fp32 scale, |x|^2} per row
  int shadow_ = 0;
  size_t sstride_ = 0;
  uint8_t *d_shadow_ = nullptr;
  float *d_sscale_ = nullptr;   // [cap_rows][2] (int8 shadow): row scale, |x|^2 of the fp32 row
  uint32_t *d_smax_ = nullptr;  // [4]: max row scale, max |x|^2 (f32 bits, atomicMax on the device), non-finite-row flag
  float s_max_ = 0.0f, n2_max_ = 0.0f;  // their host copies, refreshed by the writers
  bool s_bad_ = false;                  // a row holds inf / NaN: the shadow bounds nothing, queries take the fp32 scan
  void shadow_convert(uint32_t row_begin, uint32_t row_end);  // on wstream_
  // shadow_ == 3: int8 rows with ONE index-wide scale, for the batched int8 MFMA pass over FLOAT16 (IP / cosine) indexes
  // created with ScanTuning::shadow8 (batch_query.cpp).  Built lazily -- and incrementally -- by the first batched query
  // that finds rows it does not cover; a row that outgrows the scale, or a delete below the built prefix, re-quantises.
  // FLOAT32 indexes created with shadow8 (shadow_ == 2: per-row scales for the single-query two-stage scan) get the same
  // index-wide-scale int8 rows for THEIR batched queries, in a buffer of their own (d_s8g_f32_), built on demand.
  uint8_t *d_s8g_f32_ = nullptr;
  size_t s8g_f32_cap_rows_ = 0;
  // h8_ (round 6): FLOAT16 IP / cosine indexes WITHOUT the stored shadow -- the batched pass quantises the fp16 rows to int8 in
  // flight (h8_quant.hpp, gemm_qs_f32_kernel<.., SRC_H8>); only the four index-wide numbers of d_s8g_stats_ exist, taken under
  // the fp16 inverse scale h8_inv_bits_ (s8g_scale_ = 1 / float(inv)).  Everything else of the int8 route is shared.
  bool h8_ = false;
  uint16_t h8_inv_bits_ = 0;
  float f8_inv_ = 0.0f;  // (h8_ on a FLOAT32 index -- knob gemm_qs_f8: the fp32 inverse scale of f8_quant1)
  bool s8g_enabled() const { return d_s8g_stats_ != nullptr; }
  const uint8_t *s8g_rows() const { return shadow_ == 3 ? d_shadow_ : (h8_ ? d_rows_ : d_s8g_f32_); }
  size_t s8g_stride() const { return round_up(dim, 16); }
  uint32_t *d_s8g_stats_ = nullptr;  // {max |x_i| (f32 bits), max |x8|^2, max |ex|^2 (f32 bits), non-finite flag}
  float s8g_scale_ = 0.0f;           // the scale the built rows were quantised with (0: nothing built)
  uint32_t s8g_built_ = 0, s8g_seen_ = 0;  // rows [0, built) are quantised; rows [0, seen) went into max |x_i|
  bool ensure_shadow8g();            // true: d_shadow_ covers every row and bounds them
  // L2 indexes of FLOAT16 / BFLOAT16 rows: |x|^2 / 2 per row (fp32) for the batched matrix-core pass (batch_query.cpp,
  // gemm_qs_kernels.hip "L2").  Built lazily and incrementally by the first batched query that finds rows it does not
  // cover; a delete below the built prefix recomputes from there.  4 bytes per row.
  float *d_hnorm_ = nullptr;
  size_t hnorm_cap_rows_ = 0;
  uint32_t hn_built_ = 0;
  uint32_t *d_hn_bad_ = nullptr;  // [0] set by the kernel if a row's norm is not finite; [1] bits of the largest stored half norm
  bool hn_bad_ = false;
  float hn_max_ = 0.0f;           // upper bound of |x|^2 / 2 over the rows covered (deletes do not lower it: a bound may be loose)
  // relative error band of the pass against the exact scan, per unit of |x|^2/2 + |q|^2/2 (docs/DESIGN_NOTES.md section 3 "L2 on the
  // matrix cores"): the stored norms are shrunk by (1 - rel/2)
  // (FLOAT32 rows go through the matrix cores as bf16: twice gemm_qs_f32_rel of |x||q| <= hn + hq on top, kernels.hpp)
  float hn_rel() const {
    return (float)dim * 9.5367431640625e-07f + 1.9073486328125e-06f  // dim 2^-20 + 2^-19
           + (type == VecSimType_FLOAT32 ? 2.0f * gemm_qs_f32_rel(dim) : 0.0f);
  }
  bool ensure_half_norms();                          // true: d_hnorm_ covers every row, all finite
  float half_sq_norm_host(const void *blob) const;   // |q|^2 / 2 of a query blob of the index's type (fp32)
  bool two_stage_topk(QueryCtx *c, uint32_t n, uint32_t k, std::vector<Hit> &out);
  // the single-query path; the caller holds the shared lock and has flushed
  VecSimQueryReply *topk_locked(const void *query, size_t k, void *tctx, VecSimQueryReply_Order order);
  void topk_pass_mq(TopkJob *const *jobs, size_t n_jobs, uint32_t n);  // >= 2 jobs, shared lock held
  void topk_pass_wide(TopkJob *const *jobs, size_t n_jobs);            // 17 .. kWidePass jobs through topk_batch (no lock held)
  // multi-value top-K over one key array (key_bytes wide): false = the timeout callback fired
  bool multi_walk(QueryCtx *c, const uint32_t *d_keys, uint32_t n, size_t k, void *tctx, std::vector<VecSimQueryResult> &res);
  // the same for indexes that carry the int8 shadow: ONE multi-query pass over the shadow (scan_mq_i8_kernel), the
  // two-stage scan's error-banded filter for every query at once, survivors re-scored from the fp32 rows
  void topk_pass_mq_shadow8(TopkJob *const *jobs, size_t n_jobs, uint32_t n);  // 2 .. 8 jobs, shared lock held
  bool shadow8_mq_capable(size_t k) const;
  // normalised fp32 query -> its int8 copy, scale, |q|^2 and the error band of the shadow distance (two_stage_topk's
  // analysis); false: the band bounds nothing for this query (zero / non-finite query, non-finite rows)
  bool shadow8_query(const float *qf, int8_t *q8, float *sq, float *qn2, float *eps) const;
  // ---- the coalescer: queries that arrive while a pass is in flight join the next pass -----------------------------
  // Leader / follower: the caller at the head of the queue runs the pass for everybody queued behind it (at most
  // kMqMaxQueries), on its own thread; the others sleep until their reply is there or it is their turn to lead.  A new
  // leader waits a moment (linger) for the callers of the previous pass to come back with their next query.
  struct Coalescer {
    std::mutex mu;
    std::condition_variable cv, cv_leader;
    bool busy = false, lingering = false;
    std::deque<TopkJob *> waiting;
    uint32_t last_b = 1;  // jobs in the previous pass
  } co_;
  // rows (and their shadow) grow without copies once they are large: virtual range + mapped chunks (grow_buffer.hpp);
  // d_rows_ / d_shadow_ cache the buffers' current base addresses
  GrowBuffer rows_buf_, shadow_buf_;
  uint8_t *d_rows_ = nullptr;
  uint64_t *d_labels_ = nullptr;
  size_t cap_rows_ = 0;
  uint32_t n_rows_ = 0;  // rows resident in HBM
  // pinned staging block for AddVector
  uint8_t *h_stage_ = nullptr;
  size_t stage_cap_ = 0, stage_n_ = 0;
  hipStream_t wstream_ = nullptr;
  // host maps live in memory from the installed VecSimMemoryFunctions and are counted (memory())
  size_t host_bytes_ = 0;
  std::vector<uint64_t, HookAlloc<uint64_t>> row_label_{HookAlloc<uint64_t>(&host_bytes_)};  // committed + staged rows
  LabelTable labels_;  // label -> row(s): host copy + the device copy the kernels read (label_table.hpp)
};

// ---- reply objects (plain C structs behind the opaque ABI types) ----------------------------------
}  // namespace rsgpu

struct VecSimQueryResult {
  size_t id;
  double score;
};
struct VecSimQueryReply {
  VecSimQueryResult *results;
  size_t len;
  VecSimQueryReply_Code code;
};
struct VecSimQueryReply_Iterator {
  VecSimQueryReply *reply;
  size_t pos;
};

// the opaque ABI handle: one FLAT index on one device, or -- created with the "shards" knob -- the same interface over
// several device shards (sharded_index.hpp); exactly one of the two is set
struct RSGPU_ShardedIndex;
struct VecSimIndex {
  rsgpu::FlatIndex *flat;
  RSGPU_ShardedIndex *sharded = nullptr;
};

namespace rsgpu {
std::string &last_error();  // per-thread message behind RSGPU_LastError
// Exact k smallest composites of keys[0..n) (u32 or u64 keys) above `lower`, sorted ascending.
// Drives select_kernels.hip on c->stream and synchronises it.
void radix_select(QueryCtx *c, const void *d_keys, int key_bytes, uint32_t n, uint32_t k, const Bound &lower,
                  std::vector<Hit> &out, Bound *upper);
// frees the device buffers parked by the search seam's pool (search_abi.cpp)
void release_search_pool();
// ... and the batched path's pooled scratch (batch_query.cpp)
void release_batch_pool();
// k smallest (key,index) of a u32 key array: one-workgroup select for short arrays, radix levels otherwise
void select_keys32(QueryCtx *c, const uint32_t *d_keys, uint32_t n, uint32_t k, std::vector<Hit> &out);
// asynchronous halves of its one-sync paths (flat_index.cpp): enqueue -> [caller synchronises c->stream] -> collect
int select_keys32_enqueue(QueryCtx *c, const uint32_t *d_keys, uint32_t n, uint32_t k);
bool select_keys32_collect(QueryCtx *c, int mode, uint32_t n, uint32_t k, uint32_t *got);
VecSimQueryReply *new_reply(size_t len, VecSimQueryReply_Code code);
bool timed_out(void *timeout_ctx);

// Batch iterator: keys for all rows are computed once, every Next() selects the next-best n
// composites above the previous batch ([upstream-memory D7]).
struct BatchIterator {
  FlatIndex *index;
  QueryCtx *ctx;
  void *timeout_ctx;
  std::vector<uint8_t> query;  // copied at New (reference c_wrappers/vecsim/src/batch.rs:52-55)
  uint32_t n = 0;              // rows at creation
  uint32_t returned = 0;
  Bound lower;
  bool scanned = false;
  uint64_t epoch = 0;          // index->layout_epoch at the scan
  bool recount = false;        // rows were deleted under the iterator: `returned` no longer says how many rows lie below
                               // the bound; exhaustion is detected by a selection that comes back short
  std::unordered_set<uint64_t> seen_labels;  // multi-value: labels already yielded
};

struct AdhocCtx {
  FlatIndex *index;
  QueryCtx *ctx;
};
}  // namespace rsgpu
