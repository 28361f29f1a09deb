// kernels.hpp -- launch interface of the hand-written gfx950 kernels (scan_kernels.hip,
// select_kernels.hip).  Host code sees only these plain functions; every pointer is a device pointer
// unless noted.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace rsgpu {

// element types served by the scan kernels (values match VecSimType)
enum : int { KT_F32 = 0, KT_F64 = 1, KT_BF16 = 2, KT_F16 = 3, KT_I8 = 4, KT_U8 = 5 };
// kernel metrics: for the floating-point types cosine is IP over rows/query normalised up front;
// INT8/UINT8 rows cannot be normalised in place, their cosine divides by the two norms (KM_COS)
// KM_IPS (KT_I8 scan only): 1 - dot * row_scale[row] * query_scale, the int8-shadow filter pass
// KM_L2S (KT_I8 scan only): |q|^2 + |x|^2[row] - 2 * dot * row_scale[row] * query_scale, the same for L2 indexes
enum : int { KM_L2 = 0, KM_IP = 1, KM_COS = 2, KM_IPS = 3, KM_L2S = 4 };
// distances (and keys) of FLOAT64 indexes are 8 bytes wide, everything else computes fp32 distances
struct RowBand;
inline int key_bytes_of(int type) { return type == KT_F64 ? 8 : 4; }

// ---- label (doc id) -> storage row, as the kernels see it (label_table.hpp keeps it current under the index's writer lock) ----
// The reference looks a candidate's vector up by label (src/iterators/hybrid_reader.c:309-327, VecSimIndex_GetDistanceFrom_Unsafe);
// a label is a doc id: documents without the vector field have no row, an update is delete + a new id (src/indexer.c:179-190).
constexpr uint32_t kNoRow = 0xFFFFFFFFu;
struct LabelRows {
  const uint32_t *row_of;  // [span] first row of label base + i (>= n_rows: none); nullptr: identity labels, row = label - base
  const uint32_t *next;    // multi-value indexes off identity labelling: [n_rows] the label's next row (>= n_rows: none); else nullptr
  uint64_t base;
  uint32_t span;           // labels in [base, base + span) may have a row
  uint32_t n_rows;         // committed rows
  // round 6: labels too far apart for a direct table (1 M vectors in a 10^9-document index) -- an open-addressing hash table in HBM,
  // 16 bytes per slot {label lo, label hi, first row, used}, linear probing, at most half full (label_table.hpp HASH)
  const void *hash;        // nullptr: the forms above
  uint32_t hash_mask;      // slots - 1 (a power of two)
};
// the tables' hash of a label (splitmix64 finaliser); host and device probe alike
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint64_t label_hash(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}
#ifdef __HIPCC__
// first row of doc id `id`, kNoRow when the document has no vector
__device__ __forceinline__ uint32_t label_first_row(const LabelRows &m, uint64_t id) {
  if (m.hash) {
    typedef uint32_t lr_u4 __attribute__((ext_vector_type(4)));
    const lr_u4 *t = static_cast<const lr_u4 *>(m.hash);
    const uint32_t lo = (uint32_t)id, hi = (uint32_t)(id >> 32);
    uint32_t p = (uint32_t)label_hash(id) & m.hash_mask;
    for (uint32_t guard = 0; guard <= m.hash_mask; guard++, p = (p + 1) & m.hash_mask) {
      const lr_u4 e = t[p];
      if (!e.w) return kNoRow;
      if (e.x == lo && e.y == hi) return e.z < m.n_rows ? e.z : kNoRow;
    }
    return kNoRow;
  }
  const uint64_t off = id - m.base;
  if (id < m.base || off >= (uint64_t)m.span) return kNoRow;
  const uint32_t r = m.row_of ? m.row_of[off] : (uint32_t)off;
  return r < m.n_rows ? r : kNoRow;
}
#endif
// table maintenance (corpus_kernels.hip): dst[i] = first + (i - lo) for i in [lo, hi), kNoRow elsewhere, i in [begin, end)
void launch_label_fill(uint32_t *dst, size_t begin, size_t end, size_t lo, size_t hi, uint32_t first, hipStream_t s);
// dst[idx[i]] = val[i], i < n (duplicate indices carry the same value)
void launch_label_scatter(uint32_t *dst, const uint32_t *idx, const uint32_t *val, uint32_t n, hipStream_t s);
// dst[idx[i]] = val[i] for 16-byte slots (the hash form)
void launch_label_scatter16(void *dst, const uint32_t *idx, const void *val, uint32_t n, hipStream_t s);
// dst[i] -= 1 for i < n: an uploaded host table (0 = none, row + 1) becomes the device form (kNoRow = none)
void launch_label_decode(uint32_t *dst, size_t n, hipStream_t s);

// Orderable key of an fp32 distance: ascending key <=> ascending distance, NaN last.
// (host mirror of the device f2key in scan_kernels.hip)
inline uint32_t dist_to_key(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xFFFFFFFFu;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float key_to_dist(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

inline uint64_t dist64_to_key(double f) {
  uint64_t u;
  memcpy(&u, &f, 8);
  if ((u & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) return ~0ull;
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double key_to_dist64(uint64_t k) {
  uint64_t u = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
  double f;
  memcpy(&f, &u, 8);
  return f;
}

struct ScanTuning {
  int blocks_per_cu = 16;  // 256-thread blocks per CU the grid is sized for (profiles/r01_tune_scan_*.json)
  int rows_per_group = 0;  // 0 = per-shape default (U in the kernel)
  int nontemporal = 1;     // stream the corpus with nt loads
  int gemm_dma = 1;        // batched path: 1 LDS-DMA ring KC=8 (default), 0 register-staged, 2/3 experiments
  int filter_select = 1;   // small-K top-K: sample threshold + one filter pass (0 = radix levels only)
  int shadow16 = 0;        // FLOAT32 cosine indexes created while set keep an fp16 shadow of the rows: the scan
                           // reads the shadow, an error-bounded filter keeps the few rows that can still be in
                           // the top-K, and only those are re-scored from the fp32 rows (exact, bit-identical)
  int shadow8 = 0;         // same with an int8 shadow (+ per-row scale): a quarter of the bytes, wider error band
  int two_stage = 1;       // query-time switch of the above for indexes that carry a shadow
  int cache_decoded = 1;   // posting lists: keep the decoded id/freq arrays after the first decode (0 = decode per query)
  int coalesce_shadow8 = 1;  // coalesce concurrent K <= 16 queries on indexes that carry the int8 shadow into multi-query two-stage passes
  int mq16 = 1;            // multi-query scan: nine to sixteen FLOAT32 queries in ONE pass, queries in LDS (0 = two passes of up to
                           // eight, 2 = one pass with the queries in registers; A/B knob)
  int hybrid_tiles = 1;    // RSGPU_HybridQuery without hits_out: the query in two launches (hybrid_kernels.hip); 0 = the staged pipeline
  int hybrid_tree_tiles = 1;  // ... its general form (hybrid_tree_tile_kernel): hit list wanted, 5-8 lists, slop-dependent scorers over offsets, RSGPU_HybridTreeQuery; 0 = staged
  int prioritize_union_children = 0;  // the module's prioritizeIntersectUnionChildren (src/config.h:451, off by default): a child union's sort key in an intersection is estimate x children
  int hybrid_force_general = 0;  // diagnostics: RSGPU_HybridQuery takes the general tile kernel even for the shapes the two-launch form serves
  int hybrid_dir = 1;      // ... a probed list's window ends come from its bucket directory (one round trip; 0 = wave-wide searches)
  int hybrid_packed_docs = 1;  // document tables uploaded while set also keep {doc length, doc score} side by side (one gather per hit)
  int hybrid_select_split = 1;  // ... a tile's rank-by-count selection over up to 128 entries shares the counting among the workgroup's four wavefronts (0: the first n threads count all n)
  int hybrid_knn_pipeline = 1;  // ... the tile kernel requests the next step's vector rows before it reduces this step's distances
  int hybrid_coalesce = 1;  // ... the two-launch queries of concurrent callers share grids (hybrid_entry.hpp: the hybrid coalescer); 0 = every query its own two launches
  int hybrid_coalesce_depth = 2;  // ... grids in flight per device before arriving callers queue (1..8)
  int hybrid_coalesce_interleave = 0;  // ... 1: a shared grid deals the queries' tiles out in turn (every query's vector-bearing tiles first: measured 10-15 % SLOWER,
                                       // profiles/r06_hybrid_coalesce_ab.json); 0: one query after the other
  int hybrid_poll = 1;     // ... the host polls completion flags in pinned memory instead of synchronising the stream
  int hybrid_trace = 0;    // diagnostics: the tile kernel records a phase clock per tile (RSGPU_HybridTrace)
  int hybrid_surv_cap = 4096;  // ... candidates its reduce kernel ranks at the bound before it hands the query back (tests: small values)
  int probe_dpt = 4;       // intersection of long lists: drivers per thread of the probe / write kernels (1 = tiles of 256; A/B knob)
  int decode_lean = 1;     // the two-launch hybrid query decodes doc ids + frequencies only (a Full-codec list: 8 of 20 bytes per posting; A/B knob)
  int decode_pair = 1;     // decode-per-query mode: two qint lists of a query in one launch (A/B knob)
  int decode_dense = 1;    // qint lists without inline offsets: groups of sixteen blocks whose records are all of the minimal length decode data-parallel (decode_dense_kernel; 0: eight lanes per block everywhere)
  int decode_sync = 1;     // qint lists: the first decode leaves sub-block sync points, later decodes use 8 lanes per block (A/B knob)
  int gemm_qs = 1;         // batched path: query-stationary filter pass (gemm_qs_kernels.hip); 0 = tiled GEMM
  int gemm_qs_f32 = 2;     // FLOAT32 indexes: batched / coalesced queries through the matrix cores, rows converted to bf16 in flight
                           // (gemm_qs_f32_kernel; 0 = off: the exact multi-query scan; 2 = four waves x 64 queries -- every converted
                           // fragment feeds two MFMAs, half the LDS reads: 6.20 vs 6.56 ms per pass, profiles/r04_batch_f32_shapes_ab.json
                           // -- 1 = eight waves x 32 queries)
  int gemm_qs_h8 = 5;      // FLOAT16 IP / cosine indexes: the batched pass quantises the fp16 rows to int8 IN FLIGHT (h8_quant.hpp) and runs on the
                           // int8 matrix cores -- no stored shadow.  5 (default) = once per workgroup, register-staged (gemm_qs_h8r_kernel, 2.91 ms
                           // per configs[2] pass); 2 / 1 = in every wave, behind the LDS-DMA ring (four waves x 64 queries 3.87 ms / eight x 32
                           // 4.29 ms); 3 / 4 = 2 with smaller ring slots (4.14 / 4.61 ms); 0 = the fp16 MFMA pass (3.96 ms)
  int batch_select_regs = 1;  // batch_select_kernel keeps a list of up to 16 Ki entries in registers (0: every digit pass re-reads it -- the longer lists' path, for the tests)
  int batch_prune = 1;     // batched passes: a threshold select leaves only the candidates its new bound admits in the list (0: the lists only grow)
  int gemm_qs_f8 = 1;      // FLOAT32 IP / cosine indexes: the batched / coalesced passes quantise the fp32 rows to int8 in flight (gemm_qs_h8r_kernel<..,
                           // SRC_F8>: 5.48 ms per 256-query pass over 10M x 768, 0.70 of HBM) instead of rounding them to bf16 (gemm_qs_f32_kernel:
                           // 6.15 ms); read at index creation and at query time; 0 = the bf16 route
  int qs_phases = 0;       // batched pass: 4 = one more, shorter first filter phase (A/B knob)
  int qs_force_i8 = 0;     // timing experiment: run the query-stationary pass with the int8 MFMA over whatever bytes are there
  int vmm = 1;             // row matrices above 256 MiB grow by mapping physical chunks behind a reserved virtual range
                           // (no copy, no transient 2x HBM); 0 = hipMalloc + copy on every growth
  int vmm_chunk_mib = 0;        // 0 = automatic (256 MiB, or 1 GiB for corpora reserved large); A/B knob
  int vmm_reserve_factor = 64;  // virtual range of a mapped row matrix = factor x its size at mapping time (>= 64 GiB)
  int coalesce = 1;        // VecSimIndex_TopKQuery calls that arrive while a pass is in flight join the next pass (one
                           // multi-query scan, scan_mq_kernels.hip); replies are bit-identical to uncoalesced ones
  int coalesce_wide = 1;   // more than sixteen calls queued on an index whose batches are exact (wide_pass_capable): up to 256 of
                           // them share ONE matrix-core filter pass + exact re-scoring (batch_query.cpp); 0 = sixteen per pass
  int coalesce_wide_min = 5;  // ... and already passes of this many queued calls on plain FLOAT32 indexes (the exact multi-query scan
                              // takes 5.1 ms with eight queries and 5.8 with sixteen -- VALU-bound -- against 4.7-4.9 ms for the
                              // matrix-core pass whatever the number of queries); 17 = only what the exact scan cannot hold
  int coalesce_linger_us = -1;  // -1: automatic (8 % of the estimated pass, 50..400 us); how long a new leader waits for the
                                // callers of the previous pass to come back
  int coalesce_min_mib = 64;    // corpora below this many MiB are latency-bound: concurrent single-query streams win
  int mq_blocks_per_cu = 0;     // 0 = default: grid cap of the multi-query scan (4 per CU up to four queries, 8 beyond)
  int batch_mfma = 1;           // RSGPU_FlatIndex_TopKBatch: 0 = never the matrix-core passes (every batch through the exact
                                // multi-query scan -- bit-identical to single queries; tests and A/B)
  int shards = 0;          // > 1: VecSimIndex_New builds one index over this many device shards (sharded_index.hpp)
  int shard_exchange = 0;  // sharded handles: 0 = K-way merge on the host (every shard's winners are in pinned memory already), 1 = ONE
                           // ncclAllGather of the per-shard top-k + a merge kernel (shard_comm.cpp; one device per shard)
  int shard_replicas = 0;  // with shards: every shard holds the whole corpus, queries go round-robin
  int num_cus = 256;
};
ScanTuning &scan_tuning();

// Distances of rows [row_begin,row_end) to `query`, written as orderable keys keys[row] (u32, or u64
// for KT_F64).  rows: row-contiguous, `stride` bytes per row (multiple of 16, zero padded), query padded
// alike; for KT_I8/KT_U8 one more 16-byte chunk follows the padded query: {sum q^2 (i32/u32), |q| (f32)}.
// KM_IPS / KM_L2S: row_meta[row] = {scale, |x|^2 of the fp32 row} per row; the extra query chunk is
// {0, query_scale, |q|^2, 0} as f32 bits.
void launch_scan(const void *rows, size_t stride, uint32_t dim, int type, int metric, uint32_t row_begin,
                 uint32_t row_end, const void *query, void *keys, hipStream_t s, const float *row_meta = nullptr);

// Several queries per corpus pass (scan_mq_kernels.hip): keys[b * keys_ld + row] = the key launch_scan would write for
// queries + b * qstride, bit for bit, b < nq <= kMqMaxQueries.  fp32 / fp16 / bf16 rows, IP or L2, rows of 512 B .. 4 KiB
// in the single-query scan's 32- / 64-lane shapes; INT8 / UINT8 rows (IP, L2, KM_COS) of 128 B .. 4 KiB, every query
// followed by its extra chunk as for launch_scan (qstride >= stride + 16); FLOAT64 rows (IP, L2) up to 6 KiB with u64 keys
// (keys_ld counts keys).  false (nothing launched) for anything else.
constexpr uint32_t kMqMaxQueries = 16;
// The same for the int8 shadow of a FLOAT32 index (rows of 256 / 512 / 768 / 1024 int8 elements, row_meta = {scale, |x|^2}
// per row as for launch_scan's KM_IPS / KM_L2S; queries: nq int8 rows qstride bytes apart; qx[b] = {query scale, |q|^2}):
// keys[b * keys_ld + row] = the single scan's shadow key.  nq <= 8.
bool scan_mq_i8_supported(uint32_t stride16);
bool launch_scan_mq_i8(const void *rows, size_t stride, int metric, uint32_t row_begin, uint32_t row_end, const float *row_meta,
                       const void *queries, size_t qstride, const float *qx, uint32_t nq, uint32_t *keys, uint32_t keys_ld,
                       hipStream_t s);
bool scan_mq_supported(int type, int metric, uint32_t stride16);
bool launch_scan_mq(const void *rows, size_t stride, int type, int metric, uint32_t row_begin, uint32_t row_end,
                    const void *queries, size_t qstride, uint32_t nq, void *keys, uint32_t keys_ld, hipStream_t s);
const char *last_scan_mq_kernel_name(char *buf, size_t cap);

// name of the kernel instantiation the last full scan of this process launched (template arguments + grid)
const char *last_scan_kernel_name(char *buf, size_t cap);
int last_batch_route();  // batch_query.cpp: the filter of the last RSGPU_FlatIndex_TopKBatch call (codes: rsgpu_ext.h RSGPU_LastBatchRoute)

// Distances of the rows listed in row_ids[0..m) (0xFFFFFFFF => NaN) as values out[i] (fp32; fp64 for KT_F64).
// m_dev (optional, device memory): the actual candidate count when the host only knows the upper bound m
void launch_gather(const void *rows, size_t stride, uint32_t dim, int type, int metric, const uint32_t *row_ids,
                   uint32_t m, const void *query, void *out, hipStream_t s, const uint32_t *m_dev = nullptr);

// Synthetic corpus rows [row_begin, row_begin+n_rows) written in place: element (i,j) = Philox4x32-10(seed; first_index+i, j)
// mapped to [-1,1) (corpus_kernels.hip); padding behind dim is zeroed.
void launch_philox_rows(void *rows, size_t stride, uint32_t dim, int type, uint64_t seed, uint64_t first_index,
                        uint32_t row_begin, uint32_t n_rows, hipStream_t s);

// In-place L2 normalisation of rows [row_begin,row_end) (cosine indexes, bulk device loads).
void launch_normalize_rows(void *rows, size_t stride, uint32_t dim, int type, uint32_t row_begin, uint32_t row_end,
                           hipStream_t s);

// fp16 shadow of fp32 rows [row_begin,row_end): out row stride sstride bytes (multiple of 16, zero padded)
void launch_shadow_rows(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, void *shadow,
                        size_t sstride, hipStream_t s);
// int8 shadow of fp32 rows: shadow[r][i] = rint(x[r][i] / scale[r]), scale[r] = max|x[r]| / 127;
// meta[r] = {scale[r], |x[r]|^2}; max_bits[0] / [1] = max over rows of scale / |x|^2 (f32 bits, atomicMax);
// max_bits[2] is set when a row holds a non-finite element (the shadow cannot bound such an index)
void launch_shadow8_rows(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, void *shadow,
                         size_t sstride, float *meta, uint32_t *max_bits, hipStream_t s);
// Batched two-stage scan: cand[q*cand_cap + j].y = orderable key of the fp32 IP distance of row cand[..].x to
// queries[q] (fp32, qstride bytes apart), j < cand_count[q]; the arithmetic is the single-query scan's, bit for bit.
// tau (optional): candidates whose current (shadow) key is above tau[q] are not read; they get the last key instead.
// Only row shapes the scan runs without chunk masking (stride/16 == G*ITERS, e.g. dim 128/256/384/512/768/1024 fp32).
bool batch_rescore_supported(uint32_t stride16);
bool launch_batch_rescore(const void *rows, size_t stride, uint32_t n_rows, const void *queries, size_t qstride, void *cand,
                          const uint32_t *cand_count, uint32_t cand_cap, uint32_t n_queries, const float *tau, hipStream_t s,
                          int type = KT_F32, int metric = KM_IP,   // KT_F16: fp16 rows / queries, the fp16 scan's arithmetic;
                          const struct RowBand *band = nullptr,    // KM_L2: the L2 scan's; band: the keys are upper bounds
                          bool dense = false);  // the lists were pruned to the band (launch_batch_threshold_cand prune): candidates dealt out one per group
// hn[row] = shrink * |x|^2 / 2 (fp32) of rows [row_begin, row_end) of KT_F16 / KT_BF16 / KT_F32 rows; *bad is set if one is
// not finite
void launch_half_norm_rows(int type, const void *rows, size_t stride, uint32_t row_begin, uint32_t row_end, float shrink,
                           float *hn, uint32_t *bad, hipStream_t s);
// *out_bits = max(*out_bits, bits of v[i]), i in [begin, end): the largest non-negative float of the range
void launch_max_f32_bits(const float *v, uint32_t begin, uint32_t end, uint32_t *out_bits, hipStream_t s);
// The L2 form of the batched matrix-core pass carries a PER-ROW error band (docs/DESIGN_NOTES.md section 3 "L2 on the matrix cores"):
// the pass emits lower bounds lb of the distances, the candidate lists hold upper bounds ub = lb + band(row, q), the
// re-scoring kernel derives lb back.  band(row, q) = c1 * hnorm[row] + hq2[q]; hnorm == nullptr: no band (IP passes).
struct RowBand {
  const float *hnorm = nullptr;  // the array the pass subtracts (shrunk half norms)
  const float *hq2 = nullptr;    // [256]
  float c1 = 0.0f;
  float inv2rel = 0.0f;          // hq2[q] * inv2rel = |q|^2 / 2 (the re-scoring kernel's read-free pre-test)
};
// int8 shadow of FLOAT16 / FLOAT32 rows with ONE index-wide scale (the batched int8 MFMA pass, scan_kernels.hip "int8 shadow with
// ONE index-wide scale"): stats = {max |x_i| (f32 bits), max |x8|^2 (u32), max |ex|^2 (f32 bits), non-finite flag}
void launch_absmax_rows(int type, const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end,
                        uint32_t *stats, hipStream_t s);  // type: KT_F16 or KT_F32 rows
void launch_shadow8g_rows(int type, const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end,
                          float scale, void *shadow, size_t sstride, uint32_t *stats, hipStream_t s);
// queries (fp16 / fp32) -> int8 rows q8 + qscale[q] = scale * sq + slack[q] = twice the error band of query q
void launch_quantize_queries(int type, const void *queries, size_t qstride, uint32_t dim, uint32_t n_queries, float scale,
                             const uint32_t *stats, void *q8, size_t sstride, float *qscale, float *slack, hipStream_t s);
// rows_out[i] = cand[i].x (row ids of a candidate list, i < count[0] clamped to cap)
void launch_cand_rows(const void *cand, const uint32_t *count, uint32_t cap, uint32_t *rows_out, hipStream_t s);
// cand[i].y = orderable key of dists[i]
void launch_cand_set_keys(void *cand, const float *dists, uint32_t m, hipStream_t s);

// ---- top-K selection over keys[0..n) -------------------------------------------------------------
// Total order: composite (key, row), key u32 (key_bytes=4) or u64 (key_bytes=8).  Radix select, 8
// bits per level, most significant first; the first key_bytes levels refine the key, the last four
// the row (only needed when equal keys straddle rank K).  (lkey,lrow) is an exclusive lower bound
// (batch iterator); has_lower=0 means none.  hist is [12][256] u32, zeroed by the caller.
struct SelectBufs {
  uint32_t *hist;      // [12*256]
  uint32_t *counters;  // [0] out_count  [1] status (0 ok, 1 need more levels)
  uint32_t *out_rows;  // [cap]
  void *out_keys;      // [cap] of the key type
  uint64_t *bound;     // [2] inclusive upper bound of the selected set: key, row
};
constexpr int kSelectLevelsMax = 12;
void launch_select_pass(const void *keys, int key_bytes, uint32_t n, int pass, uint32_t k, uint64_t lkey,
                        uint32_t lrow, int has_lower, const SelectBufs &b, hipStream_t s);
// After `passes_done` levels: if the selection is exact, writes the k winners (unordered) to
// out_rows/out_keys, counters[0]=k, counters[1]=0 and the bound; otherwise counters[1]=1 only.
void launch_select_collect(const void *keys, int key_bytes, uint32_t n, int passes_done, uint32_t k, uint64_t lkey,
                           uint32_t lrow, int has_lower, const SelectBufs &b, uint32_t cap, hipStream_t s);

// ---- batched queries on the matrix cores (gemm_kernels.hip) ------------------------------------------
// S = Q[256 x K] * X^T, fp16/bf16 in, fp32 accumulate, distance = 1 - dot (IP / cosine).
// mode 0: keys_out[q*keys_ld + (row-row_begin)] = orderable key of every distance;
// mode 1: append (row,key) with distance <= tau[q] to cand[q*cand_cap + ...], counting in cand_count[q].
void launch_gemm_topk(int dtype, const void *rows, const void *queries, uint32_t stride16, uint32_t row_begin,
                      uint32_t row_end, int mode, uint32_t *keys_out, uint32_t keys_ld, const float *tau,
                      uint32_t *cand_count, void *cand, uint32_t cand_cap, hipStream_t s);
// Query-stationary FILTER pass (gemm_qs_kernels.hip): same result as launch_gemm_topk mode 1, candidates go to
// per-(workgroup, query, lane half) sub-lists sub_cand[grid][256][2][sub_cap] (row, distance bits) with
// their lengths in sub_count[grid][256][2]; launch_compact_cand concatenates them into the (row,key) lists
// launch_batch_select_cand reads.  Shapes: stride16 in {16,32,48,64,96} (dim 128..768 halves), else false.
// The rows buffer must extend 31 rows past row_end (a ragged last tile is read whole, never emitted).
bool gemm_qs_supported(uint32_t stride16);
uint32_t gemm_qs_grid(uint32_t n_rows);
bool launch_gemm_qs(int dtype, const void *rows, const void *queries, uint32_t stride16, uint32_t row_begin,
                    uint32_t row_end, const float *tau, uint32_t *sub_count, void *sub_cand, uint32_t sub_cap,
                    hipStream_t s, const float *qscale = nullptr, const float *hnorm = nullptr, const float *hq = nullptr);
// (hnorm + hq: an L2 pass over KT_F16 / KT_BF16 rows -- hnorm[row] = |x|^2 / 2 (fp32, readable up to row_end + 95), hq[q] =
// |q|^2 / 2; the candidates carry 2 (hq + hnorm - x.q) and tau bounds that)
// The same pass over FLOAT32 rows, converted to bf16 on their way from LDS to the matrix pipe (gemm_qs_f32_kernel): no
// stored shadow, HBM traffic = the fp32 rows.  stride16 = fp32 chunks per row in {32, 64, 96, 128, 192} (dim 128 .. 768);
// queries_bf16: [256] bf16 rows of 2 * dim bytes (launch_convert_queries_bf16).  A filter: |emitted - exact| <=
// gemm_qs_f32_rel(dim) * |x||q| for IP / cosine distances, twice that for L2 (hnorm / hq as for launch_gemm_qs).
bool gemm_qs_f32_supported(uint32_t stride16);
bool launch_gemm_qs_f32(const void *rows, const void *queries_bf16, uint32_t stride16, uint32_t row_begin, uint32_t row_end,
                        const float *tau, uint32_t *sub_count, void *sub_cand, uint32_t sub_cap, hipStream_t s,
                        const float *hnorm = nullptr, const float *hq = nullptr);
void launch_convert_queries_bf16(const void *queries, size_t qstride, uint32_t dim, uint32_t n_queries, void *out, size_t ostride,
                                 hipStream_t s);
// The same pass over FLOAT16 rows quantised to int8 on their way from LDS to the int8 matrix pipe (round 6; gemm_qs_f32_kernel<..,
// SRC_H8>, h8_quant.hpp): no stored shadow, HBM traffic = the fp16 rows, half the matrix-pipe cycles of the fp16 form.  stride16 =
// fp16 chunks per row in {16, 32, 48, 64, 96} (dim 128 .. 768); queries_i8 / qscale: launch_quantize_queries with scale = 1 /
// float(inv_h); inv_h_bits: the fp16 inverse scale the rows' error maxima were taken with (launch_h8_stats).
bool gemm_qs_h8_supported(uint32_t stride16);
bool launch_gemm_qs_h8(const void *rows, const void *queries_i8, uint32_t stride16, uint32_t row_begin, uint32_t row_end, const float *tau,
                       uint32_t *sub_count, void *sub_cand, uint32_t sub_cap, hipStream_t s, const float *qscale, uint16_t inv_h_bits);
// ... and over FLOAT32 rows (gemm_qs_h8r_kernel<.., SRC_F8>; stride16 = fp32 chunks per row in {32, 64, 96, 128, 192}; inv: the fp32
// inverse scale of launch_f8_stats)
bool gemm_qs_f8_supported(uint32_t stride16);
bool launch_gemm_qs_f8(const void *rows, const void *queries_i8, uint32_t stride16, uint32_t row_begin, uint32_t row_end, const float *tau,
                       uint32_t *sub_count, void *sub_cand, uint32_t sub_cap, hipStream_t s, const float *qscale, float inv);
void launch_f8_stats(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, float inv, uint32_t *stats,
                     hipStream_t s);
// stats[1] = max |x8|^2, stats[2] = max |ex|^2 (f32 bits) over FLOAT16 rows [row_begin, row_end) under THAT quantiser (atomicMax)
void launch_h8_stats(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, uint16_t inv_h_bits,
                     uint32_t *stats, hipStream_t s);
// |x~.q~ - x.q| / (|x||q|) of that pass against the exact scan: both operands rounded to bf16 (u = 2^-9, to nearest:
// 2u + u^2), the MFMA's and the scan's fp32 summation orders (dim 2^-24 each, of sum |x_i q_i| (1 + u)^2), 2 % to spare
inline float gemm_qs_f32_rel(size_t dim) { return (0.00390625f + 3.9e-6f + (float)dim * 1.2e-7f * 1.004f) * 1.02f; }
// append != 0: the sub-lists are appended behind the cand_count[q] candidates already there
void launch_compact_cand(const uint32_t *sub_count, const void *sub_cand, uint32_t sub_cap, uint32_t n_wg,
                         uint32_t *cand_count, void *cand, uint32_t cand_cap, int append, hipStream_t s,
                         const struct RowBand *band = nullptr);  // band: key = sub-list distance + band(row, q)
// per query (one workgroup each): tau_out[q] = k-th smallest distance among keys[q*ld .. +n)
// (stride > 1: element i is keys[q*ld + i*stride], a strided sample of a longer key array)
// (slack is added to every finite bound written: the two-stage scan's error band)
void launch_batch_threshold(const uint32_t *keys, uint32_t ld, uint32_t n, uint32_t k, uint32_t n_queries,
                            uint32_t n_valid, float *tau_out, hipStream_t s, uint32_t stride = 1, float slack = 0.0f,
                            const float *slack_q = nullptr);  // slack_q: per-query band (device), overrides slack
// tau_out[0] = upper bound of the k-th smallest key (k <= 1024) of keys[0..n): k-th smallest of the minima
// of 1024 groups of `per` sampled keys (per % 4 == 0, n >= 1024*per); also zeroes zero4[0..3] if given
void launch_sample_threshold(const uint32_t *keys, uint32_t n, uint32_t per, uint32_t k, float *tau_out,
                             uint32_t *zero4, hipStream_t s);
// the two kernels above for n_queries key arrays at once (keys + b * keys_ld -> tau[b], cand + b * cap, cand_count[b];
// the threshold kernel zeroes cand_count[b])
void launch_sample_threshold_batch(const uint32_t *keys, uint32_t keys_ld, uint32_t n, uint32_t per, uint32_t k,
                                   uint32_t n_queries, float *tau_out, uint32_t *cand_count, hipStream_t s);
void launch_filter_keys_batch(const uint32_t *keys, uint32_t keys_ld, uint32_t n, uint32_t n_queries, const float *tau,
                              void *cand, uint32_t *cand_count, uint32_t cap, hipStream_t s,
                              const float *slack_q = nullptr);  // slack_q[b]: added to tau[b] first (an error band per query)
// candidates of a single key array: append (row,key) of every key <= orderable(*tau) to cand[0..cap),
// counting in cand_count[0]
// (slack is added to *tau first: the two-stage scan's error bound)
void launch_filter_keys(const uint32_t *keys, uint32_t n, const float *tau, void *cand, uint32_t *cand_count,
                        uint32_t cap, hipStream_t s, float slack = 0.0f);
// per query: tau_inout[q] = k-th smallest distance among its candidates so far (kept if it has fewer than k); prune: the list
// (cand / cand_count, rewritten in place, order not kept) keeps only the candidates whose key is at or below the new bound --
// for lists whose keys are the distances the bound is compared with (NOT the upper bounds of the L2 passes)
void launch_batch_threshold_cand(const void *cand, const uint32_t *cand_count, uint32_t cand_cap, uint32_t k,
                                 uint32_t n_queries, uint32_t n_valid, float *tau_inout, uint32_t *overflow,
                                 hipStream_t s, float slack = 0.0f, const float *slack_q = nullptr, bool prune = false);
// per query: the k smallest (key,index) of keys[q*ld .. +n) -> out_rows/out_keys[q*k_ld ..], out_n[q]
void launch_batch_select_keys(const uint32_t *keys, uint32_t ld, uint32_t n, uint32_t k, uint32_t n_queries,
                              uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_n, uint32_t k_ld, hipStream_t s);
// per query: the k smallest (key,row) of its candidate list; overflow[q]=1 if the list overflowed
void launch_batch_select_cand(const void *cand, const uint32_t *cand_count, uint32_t cand_cap, uint32_t k,
                              uint32_t n_queries, uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_n,
                              uint32_t k_ld, uint32_t *overflow, hipStream_t s);

// ---- range query: all rows with key <= max_key ----------------------------------------------------
// counters[0] receives the count (collect=0) or is used as the append cursor (collect=1).
void launch_range(const void *keys, int key_bytes, uint32_t n, uint64_t max_key, int collect, uint32_t *counters,
                  uint32_t *out_rows, void *out_keys, uint32_t cap, hipStream_t s);

}  // namespace rsgpu
