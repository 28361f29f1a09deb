/*
 * rsgpu_ext.h -- NON-ABI extensions of the MI355X FLAT engine.
 *
 * Nothing here exists in VecSim; RediSearch never calls these.  They serve (a) bulk loading of
 * device-resident corpora (the 30 GB bench corpus is generated in HBM), (b) device-side results for
 * the multi-GPU top-K merge over RCCL, (c) the batched-query GEMM path the reference has no API for
 * (VecSim answers B queries with B TopKQuery calls), (d) the GPU posting-list /
 * scorer kernels whose reference counterparts sit behind Rust iterators (SURVEY.md 8b boundary 3),
 * and (e) measurement hooks for bench.py.  Plain C, pointers and sizes only.
 */
#ifndef RSGPU_EXT_H
#define RSGPU_EXT_H

#include <stddef.h>
#include <stdint.h>

#include "VecSim/vec_sim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* last error message of the calling thread ("" if none) */
const char *RSGPU_LastError(void);
/* number of usable HIP devices (0 => every VecSimIndex_New fails) */
int RSGPU_DeviceCount(void);

/* ---- FLAT index extensions -------------------------------------------------------------------- */
/* Pre-size the HBM corpus (rows). 0 on success. */
int RSGPU_FlatIndex_Reserve(VecSimIndex *index, size_t rows);
/* Append n rows that already live on the index's device, tightly packed (dim*sizeof(type) bytes
 * each), labelled first_label .. first_label+n-1. Cosine rows are normalised on the device.
 * Returns rows added or -1. */
int RSGPU_FlatIndex_AddDeviceRows(VecSimIndex *index, const void *dev_rows, size_t n, size_t first_label);
/* Append n SYNTHETIC rows generated in place in HBM (bench / test corpora; nothing crosses PCIe): element j of row i is
 * Philox4x32-10 keyed by `seed`, counter (first_index + i, j / 4), word j % 4, mapped to [-1, 1) on a 2^-23 grid
 * (FLOAT16 / BFLOAT16: that fp32 rounded to nearest even; FLOAT64: widened; INT8 / UINT8: the word's top byte).  Any
 * host can regenerate any row (oracle/flat_oracle.c oracle_philox_rows).  Labels first_label .. first_label+n-1.
 * Returns rows added or -1. */
long RSGPU_FlatIndex_AddPhiloxRows(VecSimIndex *index, uint64_t seed, uint64_t first_index, size_t n, size_t first_label);
/* Stored rows [row_begin, row_begin+n) in storage order, as they are in HBM (cosine rows: normalised), tightly packed
 * into host memory (n * dim * sizeof(type) bytes).  Inspection / tests.  0 on success. */
int RSGPU_FlatIndex_ReadRows(VecSimIndex *index, size_t row_begin, size_t n, void *host_out);
/* How the index maps labels (doc ids) to storage rows right now (csrc/label_table.hpp): 0 identity labelling -- label = base +
 * row, no table; 1 a direct-addressed table in HBM that the hybrid kernels read (it survives DeleteVector, re-adds under new
 * ids, documents without a vector, multi-value labels); 2 labels too far apart for that (1 M vectors in a 10^9-document index:
 * reference src/document.c:712-725 gives only documents with the vector field a row) -- an open-addressing hash table in HBM that
 * the same kernels read (round 6; rounds 1-5: host hash maps, host translation).  -1: a sharded handle (ask the shards).  Inspection / tests; the reference looks vectors up by
 * label in src/iterators/hybrid_reader.c:309-327. */
int RSGPU_FlatIndex_LabelTable(VecSimIndex *index);
/* Top-k of one host query written to DEVICE buffers (k fp32 scores, k u64 labels; unused slots get
 * +inf / UINT64_MAX), ordered by (score,label). The call returns after the results are complete.
 * Returns the number of hits or -1. */
int RSGPU_FlatIndex_TopKDevice(VecSimIndex *index, const void *query, size_t k, float *dev_scores,
                               uint64_t *dev_labels);
/* B queries at once: queries = [n_queries][dim] elements of the index type (host). Writes
 * ids_out/scores_out [n_queries][k] ordered by (score,label) and counts_out[n_queries]. FLOAT16 /
 * BFLOAT16 IP or COSINE indexes take the matrix-core GEMM path (256 queries per corpus pass); so do FLOAT32
 * COSINE indexes created with the "shadow16" knob (filter pass over the fp16 shadow, survivors re-scored from the
 * fp32 rows: results bit-identical to single queries); every other configuration loops over the single-query
 * path. 0 on success. */
int RSGPU_FlatIndex_TopKBatch(VecSimIndex *index, const void *queries, size_t n_queries, size_t k, size_t *ids_out,
                              double *scores_out, size_t *counts_out);
/* k best of m (score,label) candidates held in device memory (e.g. after an RCCL all-gather of
 * per-shard top-k), ascending (score,label), written to HOST arrays. `wait_stream` (hipStream_t or
 * NULL) is synchronised first. Returns the number written or -1. */
int RSGPU_MergeTopK(int device, const float *dev_scores, const uint64_t *dev_labels, size_t m, size_t k,
                    double *scores_out, uint64_t *labels_out, void *wait_stream);

/* The same merge over HOST arrays (pure host code; what runs after a collective's result has been brought to the host,
 * and what the world-size-2 gloo tests exercise on CPU). */
int RSGPU_MergeTopKHost(const float *scores, const uint64_t *labels, size_t m, size_t k, double *scores_out,
                        uint64_t *labels_out);

/* ---- the exchange over RCCL, in C (shard_comm.cpp; SURVEY.md 8e, BASELINE north star) -----------------------------
 * One rank per GPU, every rank holds a row shard behind an ordinary VecSim handle.  RSGPU_ShardComm_TopK: the local shard
 * answers through the single-query path, its k winners travel as {label, orderable distance key} entries in ONE
 * ncclAllGather over xGMI, a merge kernel ranks the world x k candidates by (distance, label) on every rank and writes the
 * k best into pinned host memory -- every rank returns the global answer (the collective analogue of the coordinator's
 * per-shard top-K -> heap merge, reference src/module.c:3541-3547).  Every rank must call it with the same query and k, in
 * the same order.  Bootstrap as with any NCCL program: rank 0 calls RSGPU_ShardComm_GetUniqueId (128 bytes), the launcher
 * hands the id to every rank (MPI_Bcast, a torch.distributed broadcast, a file ...), every rank calls RSGPU_ShardComm_Init.
 * RCCL is bound at first use (dlopen); without it these return an error and RSGPU_LastError says why. */
typedef struct RSGPU_ShardComm RSGPU_ShardComm;
int RSGPU_ShardComm_GetUniqueId(void *id128);
RSGPU_ShardComm *RSGPU_ShardComm_Init(int rank, int world, const void *id128, int device);
void RSGPU_ShardComm_Free(RSGPU_ShardComm *c);
int RSGPU_ShardComm_World(const RSGPU_ShardComm *c);
/* returns the number of results written (<= k), -1 on error */
long RSGPU_ShardComm_TopK(RSGPU_ShardComm *c, VecSimIndex *local, const void *query, size_t k, uint64_t *labels_out,
                          double *scores_out);
/* out[0] exchanges, out[1] nanoseconds spent in them (H2D of the local winners + all-gather + merge kernel + sync) */
void RSGPU_ShardComm_GetStats(RSGPU_ShardComm *c, uint64_t out[2], int reset);
/* the merge kernel alone on n <= 8192 gathered candidates given on the host (labels[i] == UINT64_MAX: padding): the k best
 * by (score, label) ascending, as RSGPU_MergeTopKHost orders them.  Returns the number written or -1. */
long RSGPU_MergeTopKDevice(int device, const float *scores, const uint64_t *labels, size_t n, size_t k, double *scores_out,
                           uint64_t *labels_out);

/* ---- one index over several GPUs of one process (sharded_index.cpp) ------------------------------
 * The in-process form of the coordinator's per-shard top-K -> heap merge (reference src/module.c:3541-3547): the corpus
 * is row-partitioned over n_shards FLAT shards, shard i resident on devices[i] (NULL: i mod the visible devices;
 * several shards may share a device), a query runs on every shard concurrently and the per-shard top-k lists are merged
 * by (score, label).  replicas != 0: every shard holds the WHOLE corpus and a query goes to one of them round-robin
 * (throughput scaling for concurrent callers).  Results equal those of one unsharded index over the same vectors. */
typedef struct RSGPU_ShardedIndex RSGPU_ShardedIndex;
RSGPU_ShardedIndex *RSGPU_ShardedIndex_New(const VecSimParams *params, int n_shards, const int *devices, int replicas);
void RSGPU_ShardedIndex_Free(RSGPU_ShardedIndex *index);
int RSGPU_ShardedIndex_NumShards(RSGPU_ShardedIndex *index);
int RSGPU_ShardedIndex_ShardDevice(RSGPU_ShardedIndex *index, int shard);
/* borrowed handle of one shard, for bulk loads (RSGPU_FlatIndex_AddDeviceRows / _AddPhiloxRows) and inspection; the
 * caller keeps labels disjoint across shards */
VecSimIndex *RSGPU_ShardedIndex_Shard(RSGPU_ShardedIndex *index, int shard);
/* Drop-in form: after RSGPU_SetTuning("shards", N) (and "shard_replicas", 0 | 1), VecSimIndex_New itself returns a handle
 * over N device shards and the WHOLE VecSim C ABI works on it unchanged -- AddVector / DeleteVector (routed to the owning
 * shard), TopKQuery / RangeQuery (fan-out + merge), VecSimBatchIterator_* (per-shard iterators with look-ahead buffers
 * merged by (score, label)), VecSimIndex_AdhocBfCtx_* and GetDistanceFrom_Unsafe (labels routed to their shards),
 * PreferAdHocSearch / info over the summed sizes -- so the reference's hybrid reader and vector_index.c run on several
 * GPUs without a source change.  This returns the shards behind such a handle (NULL for a single-device handle); the
 * handle owns them. */
RSGPU_ShardedIndex *RSGPU_ShardedIndex_FromHandle(VecSimIndex *index);
size_t RSGPU_ShardedIndex_IndexSize(RSGPU_ShardedIndex *index);
/* The exchange step of the fan-out top-k queries since the last reset: out[0] queries merged, out[1] nanoseconds between
 * "the last shard's winners are in (pinned) host memory" and "the merged reply exists" -- every shard's last kernel writes
 * its K winners straight into host memory, so the exchange is a K-way host merge of N * K pairs (no collective inside one
 * process; between processes the same lists travel by one RCCL all-gather, redisearch_amd/sharded.py). */
void RSGPU_ShardedIndex_GetExchangeStats(RSGPU_ShardedIndex *index, uint64_t out[2], int reset);
/* ... and of the queries that took the RCCL exchange instead (RSGPU_SetTuning("shard_exchange", 1): one ncclAllGather of the
 * per-shard top-k + a merge kernel, shard_comm.cpp; needs one device per shard): out[0] queries, out[1] nanoseconds of the
 * exchange (H2D of the winners + all-gather + merge kernel + sync; the first one includes creating the communicators),
 * out[2] ranks of the communicator (0 before the first such query).
 * Concurrency: collectives of one communicator are ORDERED -- every rank must issue them in the same sequence -- so queries that
 * take the RCCL exchange pass it ONE AT A TIME (a mutex around all-gather + merge, shard_comm.cpp): concurrent callers still
 * overlap their shard scans (and share passes through the coalescer), only the ~tens-of-microseconds exchange serialises.  The
 * host merge (shard_exchange 0) has no such step.  Scores travel as the 64-bit orderable key of the reply's double, so FLOAT64
 * shards merge in the host merge's order. */
void RSGPU_ShardedIndex_GetRcclStats(RSGPU_ShardedIndex *si, uint64_t out[3], int reset);
/* VecSimIndex_AddVector / _DeleteVector / _GetDistanceFrom_Unsafe / _TopKQuery / _RangeQuery semantics over the
 * whole index; a label lives on exactly one shard (new labels go to the emptiest one) */
int RSGPU_ShardedIndex_AddVector(RSGPU_ShardedIndex *index, const void *blob, size_t label);
int RSGPU_ShardedIndex_DeleteVector(RSGPU_ShardedIndex *index, size_t label);
double RSGPU_ShardedIndex_GetDistanceFrom(RSGPU_ShardedIndex *index, size_t label, const void *normalized_blob);
VecSimQueryReply *RSGPU_ShardedIndex_TopKQuery(RSGPU_ShardedIndex *index, const void *queryBlob, size_t k,
                                               VecSimQueryParams *queryParams, VecSimQueryReply_Order order);
VecSimQueryReply *RSGPU_ShardedIndex_RangeQuery(RSGPU_ShardedIndex *index, const void *queryBlob, double radius,
                                                VecSimQueryParams *queryParams, VecSimQueryReply_Order order);

/* ---- measurement ------------------------------------------------------------------------------ */
/* When on, every FLAT scan launch is bracketed by HIP events on its own stream. */
void RSGPU_SetProfiling(int on);
void RSGPU_ResetProfile(void);
/* launches, summed kernel milliseconds, algorithmic bytes (rows*dim*sizeof(type)) */
void RSGPU_GetScanProfile(uint64_t *launches, double *total_ms, uint64_t *bytes);
/* The query coalescer behind VecSimIndex_TopKQuery (docs/DESIGN_NOTES.md "the coalescer"): calls that arrive while a pass over the corpus
 * is in flight join the next pass, which scores every row against all of them at once (scan_mq_kernels.hip, up to 8
 * queries per pass); replies are bit-identical to uncoalesced ones.  out[0] passes, [1] queries served, [2] passes that
 * ran the multi-query scan, [3] queries those served, [4] times a new leader waited for the previous pass's callers,
 * [5] nanoseconds spent so, [6] device nanoseconds of the multi-query scans (HIP events), [7] queries of a multi-query
 * pass whose batched selection overflowed and was redone through the radix levels. */
void RSGPU_GetCoalesceStats(uint64_t out[8]);
/* Which filter the last RSGPU_FlatIndex_TopKBatch call (of any thread; wide coalesced passes included) ran its corpus passes through:
 * 0 none yet, 1 fp16 / bf16 matrix-core pass (gemm_qs_kernel), 2 FLOAT32 rows rounded to bf16 in flight (gemm_qs_f32_kernel), 3 stored
 * fp16 shadow, 4 stored int8 shadow, 5 FLOAT16 / BFLOAT16 rows quantised to int8 in flight, 6 FLOAT32 rows quantised to int8 in flight
 * (gemm_qs_h8r_kernel), 7 no matrix-core pass (small corpus / K above 1024: exact multi-query scans), 8 the L2 form of 1.  Every route
 * re-scores its survivors exactly; this is a record for benches and tests that claim a route, not a contract. */
int RSGPU_LastBatchRoute(void);
void RSGPU_ResetCoalesceStats(void);
/* ... of those passes, the WIDE ones (round 4): more than sixteen calls queued on an index whose batched queries are exact
 * (FLOAT32 cosine / L2 with or without a shadow, FLOAT16 / BFLOAT16 L2) share ONE matrix-core filter pass + exact
 * re-scoring, up to 256 per pass, replies bit-identical to serial ones.  out[0] wide passes, [1] queries they served.
 * Knob "coalesce_wide" (default 1).  Reset by RSGPU_ResetCoalesceStats. */
void RSGPU_GetWidePassStats(uint64_t out[2]);
/* calls that left the coalescer's queue because their timeout callback fired while they waited (round 4): a queued call
 * polls ITS OWN timeoutCtx on ITS OWN thread every millisecond -- the callback never runs on another caller's thread -- and
 * returns VecSim_QueryReply_TimedOut without waiting for the pass in flight.  Reset by RSGPU_ResetCoalesceStats. */
uint64_t RSGPU_GetCoalesceTimeouts(void);
/* like RSGPU_GetLastScanKernel, for the multi-query scan */
const char *RSGPU_GetLastMqScanKernel(char *buf, size_t cap);
/* Two-stage (shadow) scans of this process since the last reset: out[0] attempts, [1] answered by the two-stage path,
 * then the ways out to the plain fp32 scan (exact, but 4x the bytes): [2] unsupported shape, [3] query / error band not
 * finite (zero query, non-finite row), [4] the first (sampled-bound) pass overflowed the candidate buffer, [5] the rows
 * inside the error band overflowed it, [6] fewer than k rows inside the band, [7] the final select overflowed. */
void RSGPU_GetTwoStageStats(uint64_t out[8]);
void RSGPU_ResetTwoStageStats(void);
/* The kernel instantiation the last full FLAT scan of this process launched, e.g.
 * "scan_kernel<f32,IP,G=64,ITERS=3,U=8,EXACT=1,NT=1> grid=4096x256" (bench.py's roofline.kernel). Returns buf. */
const char *RSGPU_GetLastScanKernel(char *buf, size_t cap);
/* Engine knobs (A/B experiments and opt-in modes). Returns 0 if the key is known.
 *   "blocks_per_cu", "rows_per_group", "nontemporal"  launch shape of the FLAT scan kernel
 *   "filter_select"  1 (default): K <= 32 uses sample threshold + one filter pass; 0: radix levels only
 *   "gemm_dma", "gemm_qs"  batched path: staging variant of the tiled GEMM; 1 (default) query-stationary filter
 *                    pass with 8 waves x 32 queries, 2 = 4 waves x 64 queries, 0 = tiled GEMM filter
 *   "cache_decoded"  1 (default): a posting list is decoded once and the decoded arrays are kept in HBM
 *   "shadow16"       0 (default); 1: FLOAT32 cosine indexes created from now on keep an fp16 shadow and answer
 *                    K <= 128 queries with the two-stage exact scan
 *   "shadow8"        same with an int8 shadow + one fp32 scale per row (a quarter of the fp32 bytes; K <= 32)
 *   "two_stage"      1 (default): query-time switch of the above for indexes that carry a shadow
 *   "shards"         0 (default); N > 1: VecSimIndex_New builds ONE index over N device shards (shard i on device
 *                    i mod the visible devices) behind the ordinary handle -- see RSGPU_ShardedIndex_FromHandle
 *   "shard_replicas" with "shards": every shard holds the whole corpus, queries go to one of them round-robin
 *   "coalesce"       1 (default): concurrent VecSimIndex_TopKQuery calls on one index share corpus passes (see
 *                    RSGPU_GetCoalesceStats); 0: every call scans on its own stream
 *   "coalesce_wide"  1 (default): more than sixteen queued calls share one matrix-core pass (RSGPU_GetWidePassStats)
 *   "gemm_qs_f32"    2 (default; 1: the eight-wave shape): FLOAT32 indexes answer batches / wide passes on the matrix cores, rows rounded to bf16 in
 *                    flight + exact re-scoring (no stored shadow); 0: the exact multi-query scan, sixteen per pass
 *   "coalesce_linger_us"  -1 (default: 5 % of a pass, 20..300 us); "coalesce_min_mib" 64: smaller corpora never coalesce
 *   "vmm"            1 (default): row matrices above 256 MiB grow by mapping physical chunks behind a reserved virtual
 *                    range (no copy, no transient 2x HBM); 0: hipMalloc + full copy on every growth */
int RSGPU_SetTuning(const char *key, int value);
/* frees idle per-query workspaces */
void RSGPU_ReleaseWorkspaces(void);

#ifdef __cplusplus
}
#endif
#endif /* RSGPU_EXT_H */
