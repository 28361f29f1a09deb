// search_kernels.hpp -- launch interface of postings_kernels.hip (posting decode, intersection,
// scorers).  Device pointers unless noted.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "kernels.hpp"

namespace rsgpu {

// record layout of a codec: qint arity and slot of each field (-1 = absent); kind 0 qint, 1 varint
// delta, 2 raw u32 delta from the block's first doc id
// wide != 0: a varint-coded field mask of up to 128 bits follows the qint fields (or the varint delta) -- the *Wide
// codecs of reference inverted_index/src/codec/{full,freqs_fields,fields_only,fields_offsets}.rs
struct CodecDesc {
  int kind, n, freq, mask, osz, wide;
};
CodecDesc codec_desc(int codec);

// One thread per IndexBlock, sequential inside the block (records are variable-length and
// delta-coded), blocks in parallel.  ids/freqs/masks receive entry_off[b] + e.
// Optional outputs (NULL = skipped): wmasks[4*e .. 4*e+3] the 128-bit field mask (little-endian words) of wide codecs;
// off_pos[e] / off_len[e] byte position (into `bytes`) and length of the record's offsets blob.
void launch_decode_blocks(const CodecDesc &cd, const uint8_t *bytes, const uint64_t *byte_off, const uint32_t *first,
                          const uint32_t *nent, const uint32_t *entry_off, uint32_t n_blocks, uint32_t *ids,
                          uint32_t *freqs, uint32_t *masks, hipStream_t s, uint32_t *wmasks = nullptr,
                          uint32_t *off_pos = nullptr, uint32_t *off_len = nullptr, uint32_t *sync = nullptr, int sync_mode = 0,
                          uint32_t sync_span = 0,   // sync_span: the widest byte range (16-byte aligned start) of
                                                    // decode_sync_blocks_per_wave() consecutive blocks, 0 = unknown
                          uint32_t avg_block_bytes = 0);  // encoded bytes / blocks: sizes the lane-per-block modes' wavefronts
// Two qint lists (kind 0, no wide masks) in ONE launch -- the same arguments as launch_decode_blocks, per list; false
// (nothing launched) when a list is not of that kind or is empty.
struct DecodeListArgs {
  CodecDesc cd;
  const uint8_t *bytes;
  const uint64_t *byte_off;
  const uint32_t *first, *nent, *entry_off;
  uint32_t n_blocks;
  uint32_t *ids, *freqs, *masks, *wmasks, *off_pos, *off_len, *sync;
  int sync_mode;
  uint32_t sync_span;
};
bool launch_decode_blocks_pair(const DecodeListArgs &a, const DecodeListArgs &b, hipStream_t s);
// Sub-block sync points of the qint layouts (decode_sync_words(n_blocks) u32 per list): sync_mode 1 = this decode writes
// them, 2 = they are valid and eight lanes share a block.  See postings_kernels.hip "sync".
bool decode_sync_supported(const CodecDesc &cd);
size_t decode_sync_words(uint32_t n_blocks);
uint32_t decode_sync_blocks_per_wave();
uint32_t decode_stage_bytes();  // bytes of encoded input a decode wavefront can stage in LDS

constexpr int kMaxLists = 32;  // children of one intersection / union (the reference's own tests go to 25)
constexpr int kMaxNodes = 64;      // nodes of one query tree (terms + aggregates; <= kMaxLists terms)
constexpr int kMaxTreeDepth = 16;  // nesting levels below the root
struct ListView {
  const uint32_t *ids[kMaxLists];
  const uint32_t *freqs[kMaxLists];
  uint32_t len[kMaxLists];
  // 64-bit doc ids: a list stores ids relative to its own base; add[l] = base_l - B, B = the smallest doc id among the
  // lists of the query, so ids[l][i] + add[l] is the document's id in the frame the lists share (add may be negative;
  // the sum fits 32 bits: the caller checks that the lists span fewer than 2^32 ids)
  long long add[kMaxLists];
  int n;  // lists; list 0 drives (the shortest)
};
// How the term ("leaf") columns of a hit list hang off the lists of a ListView: a list is a term's posting list (one
// leaf, its own freq column, entry index = position) or the hit list of a nested union / intersection (several
// leaves, each with a freq column and an entry-index column indexed by the position in that hit list).
struct LeafMap {
  int n_leaves;
  uint8_t leaf_list[kMaxLists];          // ListView slot the leaf belongs to
  const uint32_t *leaf_freq[kMaxLists];  // [position in that list] -> frequency (NULL: 0)
  const uint32_t *leaf_epos[kMaxLists];  // [position in that list] -> entry index in the term's posting list
                                         // (NULL: the position itself; 0xFFFFFFFF: the leaf did not match)
};
// probe: for every element of list 0, binary-search the other lists; flags[i]=1 on consensus,
// pos[(l-1)*len0 + i] = match position in list l; block_counts[b] = hits in block b
// dpt: drivers per thread -- 1 (tiles of 256 drivers, one block count each) or 4 (tiles of 1 024); probe and write must agree
void launch_intersect_probe(const ListView &v, uint8_t *flags, uint32_t *pos, uint32_t *block_counts, hipStream_t s, int dpt = 1);
// exclusive scan of block_counts[0..nb) in place, total -> total_out[0]
void launch_scan_counts(uint32_t *block_counts, uint32_t nb, uint32_t *total_out, hipStream_t s);
// ordered compaction: out_ids[h], out_freqs[leaf*cap + h]; out_epos[leaf*cap + h] (optional) = the hit's entry index in
// the leaf's posting list
void launch_intersect_write(const ListView &v, const LeafMap &m, const uint8_t *flags, const uint32_t *pos,
                            const uint32_t *block_off, uint32_t *out_ids, uint32_t *out_freqs, uint32_t cap, hipStream_t s,
                            uint32_t *out_epos = nullptr, int dpt = 1);

// ---- proximity over the term offsets (reference index_result/src/core/proximity.rs, index_result.c:51-103) ----------
// Leaves are the hit list's term columns; a child of the aggregate is one leaf, or -- is_agg -- a union / intersection
// of consecutive leaves whose positions are merged.  bytes/off_pos/off_len: per leaf, the list's byte buffer and the
// decoded offsets index (NULL off_pos: the codec stores no offsets).
struct OffsetView {
  const uint8_t *bytes[kMaxLists];
  const uint32_t *off_pos[kMaxLists];
  const uint32_t *off_len[kMaxLists];
};
struct ProxParams {
  int n_children, n_leaves;
  uint8_t child_first[kMaxLists + 1];
  uint8_t is_agg[kMaxLists];
  int max_slop;  // < 0: no slop constraint
  int in_order;
  int count_present;  // the aggregate is a UNION: children that did not match this document are not part of it
};
// filter the candidates of the probe: flags[i] &= within_range(candidate i); block_counts recomputed.
// candidate i's entry index is i in leaf 0 and pos[(l-1)*n0 + i] in leaf l.
void launch_prox_filter(const ProxParams &p, const OffsetView &o, const LeafMap &m, uint32_t n0, const uint32_t *pos,
                        uint8_t *flags, uint32_t *block_counts, hipStream_t s);
// slops[h] = IndexResult_MinOffsetDelta of hit h (entry indices epos[l*cap + h], 0xFFFFFFFF = leaf absent)
void launch_prox_slop(const ProxParams &p, const OffsetView &o, const uint32_t *epos, uint32_t len, uint32_t cap,
                      int32_t *slops, hipStream_t s);

// union: the lists' emitted-prefix arrays ([len+1] each); see postings_kernels.hip "union / NOT"
struct UnionView {
  const uint32_t *prefix[kMaxLists];
};
void launch_union_flag(const ListView &v, int s, uint8_t *flags, uint32_t *block_counts, hipStream_t st);
void launch_union_prefix(const uint8_t *flags, uint32_t len, const uint32_t *block_off, const uint32_t *total,
                         uint32_t *prefix, hipStream_t st);
// out_epos (optional): entry index per leaf, 0xFFFFFFFF where the leaf's list does not hold the document
void launch_union_write(const ListView &v, const LeafMap &m, const UnionView &u, int s, const uint8_t *flags,
                        uint32_t *out_ids, uint32_t *out_freqs, uint32_t cap, hipStream_t st, uint32_t *out_epos = nullptr);
// NOT: doc ids 1..max_doc (or the entries of `universe`) the child does not hold, freq 1 (virtual results)
void launch_not_range(const uint32_t *child, uint32_t child_len, uint32_t max_doc, uint32_t *out_ids,
                      uint32_t *out_freqs, uint32_t cap, hipStream_t st);
// out[0] = number of entries of the sorted list below x
void launch_count_below(const uint32_t *list, uint32_t len, uint64_t x, uint32_t *out, hipStream_t st);
// child_shift: (the universe's base) - (the child's base): a universe id u is child id u + child_shift
void launch_not_universe_flag(const uint32_t *universe, uint32_t n_u, const uint32_t *child, uint32_t child_len,
                              long long child_shift, uint32_t max_doc, uint8_t *flags, uint32_t *block_counts,
                              hipStream_t st);
void launch_not_universe_write(const uint32_t *universe, uint32_t n_u, const uint8_t *flags, const uint32_t *block_off,
                               uint32_t *out_ids, uint32_t *out_freqs, uint32_t cap, hipStream_t st);

struct ScoreParams {
  int scorer, n_lists;
  double avg_doc_len, root_weight, min_score, inv_tanh;
  double idf[kMaxLists], bm25_idf[kMaxLists], weight[kMaxLists];
  // result tree: the root (intersection, or union when is_union) has n_groups children; child g is leaf
  // group_first[g] alone (group_op 0) or a union (1) / intersection (2) of the leaves [group_first[g], group_first[g+1])
  // with weight group_weight[g].  Flat hit lists: one term group per leaf.
  int n_groups;
  uint8_t group_first[kMaxLists + 1];
  uint8_t group_op[kMaxLists];
  double group_weight[kMaxLists];
  // Deeper trees (RSGPU_EvalTreeNodes): the whole result tree in POST-ORDER over the leaf columns, n_nodes > 0.  Node i is
  // a term (op 0: leaf column node_leaf[i]) or an aggregate (1 union / 2 intersection) whose children are the complete
  // subtrees right before it; node_depth[i] = distance from the root (the root, last, has depth 0 and its weight is
  // root_weight); node_in_union[i] != 0: the node's PARENT is a union (DISMAX takes the maximum there).  The groups above
  // still describe the root's children (slop, proximity).  Depth <= kMaxTreeDepth.
  int n_nodes;
  uint8_t node_op[kMaxNodes], node_depth[kMaxNodes], node_leaf[kMaxNodes], node_in_union[kMaxNodes];
  double node_weight[kMaxNodes];
  int slop;  // IndexResult_MinOffsetDelta of offset-less children: max(n_groups-1, 1)
  const int32_t *slops;  // per-hit slop computed from the term offsets (launch_prox_slop); NULL: the constant above
  int is_union;  // hits come from RSGPU_Union: per-hit slop from the matched children, DISMAX takes the maximum
  long long table_off;  // doc-table entry of hit id x = x + table_off (64-bit doc ids: hits base - table first id)
};
// scores[h] (fp64) and keys[h] = descending-score orderable u64 (for the top-N select)
void launch_score(const ScoreParams &p, const uint32_t *ids, const uint32_t *freqs, uint32_t len, uint32_t cap,
                  const uint32_t *doc_len, const float *doc_score, const uint32_t *max_freq, uint32_t table_n,
                  double *scores, uint64_t *keys, hipStream_t s, uint32_t *keys32 = nullptr);
// keys32 (optional): orderable image of (float)(-score), a monotone 4-byte prefilter key for the top-N
// (row, full key) of the first min(*count, cap) prefilter candidates -> host-visible buffers; out_n[0] = *count
void launch_fetch_cand64(const void *cand, const uint32_t *count, uint32_t cap, const uint64_t *keys64,
                         const uint32_t *ids, uint32_t *out_rows, uint64_t *out_keys, uint32_t *out_ids, uint32_t *out_n,
                         hipStream_t s);
// out[i] = src[idx[i]], i < n (idx / out may be pinned host memory)
void launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t *out, hipStream_t s);
// the same for an index list whose length the device knows: out[i] = src[idx[i]] for i < min(*count, cap), indices
// >= src_len are skipped (idx / count / out may be pinned host memory written by the previous kernel of the stream)
void launch_gather_u32_counted(const uint32_t *src, uint32_t src_len, const uint32_t *idx, const uint32_t *count,
                               uint32_t cap, uint32_t *out, hipStream_t s);

// per-hit term records of hits [first, first+count) for one leaf: out = 7 planes of `count` words -- entry index in the
// leaf's posting list (0xFFFFFFFF absent), field mask words 0..3, offsets position / length (postings_kernels.hip)
void launch_hit_records(const uint32_t *hit_ids, uint32_t first, uint32_t count, const uint32_t *hit_epos,
                        const uint32_t *list_ids, uint32_t list_len, long long shift, const uint32_t *masks,
                        const uint32_t *wmasks, const uint32_t *off_pos, const uint32_t *off_len, uint32_t *out, hipStream_t s);

// BM25STD.NORM epilogue: scores[i] /= max(0, max_i scores[i]) unless that maximum is 0; keys rewritten alike.
// max_key_zeroed: one u64 of scratch, zeroed by the caller on the same stream
void launch_score_max_normalize(double *scores, uint64_t *keys, uint32_t len, uint64_t *max_key_zeroed, hipStream_t s);

// rows[i] = first row of doc id ids_base + ids[i] (kernels.hpp LabelRows: identity arithmetic or the device label table), else 0xFFFFFFFF
void launch_labels_to_rows(const uint32_t *ids, uint32_t n, uint64_t ids_base, const LabelRows &L, uint32_t *rows, hipStream_t s);
// the hits with a vector, compacted in any order: rows_out[slot] = row, cand[slot] = (hit index, 0) (uint2), count[0] += 1
// per hit; slots >= cap are dropped (count[0] > cap tells).  count must be zeroed by the caller on the same stream.
void launch_labels_to_cand(const uint32_t *ids, uint32_t n, uint64_t ids_base, const LabelRows &L, uint32_t *rows_out, void *cand,
                           uint32_t *count, uint32_t cap, hipStream_t s);
// multi-value indexes off identity labelling: dists[i] = min(dists[i], distances of the further rows of rows[i]'s label) for
// i < m (and < *m_dev when given) -- the arithmetic of the gather (hybrid_kernels.hip).  false: no such kernel for the type /
// metric / row length (the caller expands the labels on the host)
bool knn_chain_supported(int type, int metric, uint32_t stride16);
void launch_knn_chain_min(const void *rows, size_t stride, int type, int metric, const uint32_t *first_rows, uint32_t m,
                          const uint32_t *m_dev, const LabelRows &L, const void *query, float *dists, hipStream_t s);
// k (<= knn_topk_max_k()) best of the compacted candidates by (key of dists[slot], hit index cand[slot].x), one launch:
// winners' hit indices / u32 keys / doc ids (ids[hit]) and their number go to out_* (pinned host memory), *overflow is
// set when *count > cap.  `part`: knn_topk_scratch_bytes() of device scratch; `count` and `ticket` (device, zero on
// entry) are zero again when the kernel ends.
uint32_t knn_topk_max_k();
size_t knn_topk_scratch_bytes();
void launch_knn_topk(const float *dists, const void *cand, uint32_t *count, uint32_t cap, uint32_t k, const uint32_t *ids,
                     void *part, uint32_t *ticket, uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_ids, uint32_t *out_n,
                     uint32_t *overflow, hipStream_t s);
// keys[i] = orderable(dists[i]) (NaN last)
void launch_dist_to_keys(const float *dists, uint32_t n, uint32_t *keys, hipStream_t s);

// ---- a whole hybrid query in two launches (hybrid_kernels.hip) ---------------------------------------------------
// flat AND of up to kHybMaxLists term lists (list 0 drives) -> top_n by score next to the k nearest among the hits that have a
// vector (L: doc id -> row, kernels.hpp LabelRows).  Either branch may be off (top_n == 0 / k == 0).
// kHybMaxK: the tile's and the reduce kernel's bounds are the k-th of 64 bests (one wavefront ranks them) -- exact up to 64, tight
// up to ~32; between, more entries pass them (the reduce kernel hands a query with too many back to the exact select)
constexpr int kHybMaxLists = 4, kHybMaxK = 64, kHybMaxChunks = 512;
constexpr int kHybTracePhases = 9;  // start | window ends | window staged | probe done | hits compacted | scored | ranked | distances | end
struct HybridTileArgs {
  int n;                               // lists
  const uint32_t *ids[kHybMaxLists];   // decoded doc ids (relative to the list's base)
  const uint32_t *freq[kHybMaxLists];  // decoded frequencies (NULL: the codec stores none -- 1)
  uint32_t len[kHybMaxLists];
  long long add[kHybMaxLists];         // ListView::add
  // scoring: the flat result tree of the lists in this order; P.slops must be NULL
  uint32_t top_n;
  ScoreParams P;
  const uint32_t *doc_len;
  const float *doc_score;
  const uint32_t *max_freq;
  uint32_t table_n;
  // KNN
  uint32_t k;
  const void *rows;                    // the index's row matrix
  uint32_t stride16, chunks;           // 16-byte chunks per row (stride) / used (the same for padded rows)
  int G, ITERS;                        // pick_shape(stride16): the lanes-per-row shape of scan_kernel
  const void *query;                   // [chunks] 16-byte chunks, prepared as for the scan
  uint64_t ids_base;                   // doc id = ids_base + (ids[0][i] + add[0])
  LabelRows L;                         // doc id -> row of the index (identity arithmetic, or the device label table)
  // per tile, fixed slots
  uint32_t *tile_hits;                 // [n_tiles]
  uint64_t *part_skey;                 // [n_tiles][top_n]  ~d2key(score), ~0 = none
  uint32_t *part_sidx;                 // [n_tiles][top_n]  doc id in the shared frame (ids[0][i] + add[0])
  uint64_t *part_knn;                  // [n_tiles][k]      (distance key << 32) | doc id in the shared frame, ~0 = none
  uint32_t pool_words;                 // set by the launcher: u32 words of LDS before the query
  uint64_t *trace;                     // NULL, or [n_tiles][kHybTracePhases] clock readings (diagnostics)
  // round 4 ---------------------------------------------------------------------------------------------------------
  // bucket directory of list l >= 1 (NULL: the wave-wide searches): dir[l][b] = lower_bound(ids[l], b << dir_shift[l]) for
  // b < dir_n[l], dir[l][dir_n[l] - 1] = len[l] -- a tile's window ends are two independent loads, one round trip, where
  // the 64-ary searches were three to four dependent ones
  const uint32_t *dir[kHybMaxLists];
  uint32_t dir_shift[kHybMaxLists], dir_n[kHybMaxLists];
  const uint2 *len_score;              // NULL, or {doc length, doc score bits} per document: one 8-byte gather per hit for two
  int knn_pipeline;                    // the next step's rows are requested before this step's distances are reduced
};
// dir[b] = lower_bound(ids, b << shift), b < dir_n (ids ascending, n > 0; dir_n >= (ids[n - 1] >> shift) + 2)
void launch_build_bucket_dir(const uint32_t *ids, uint32_t n, uint32_t shift, uint32_t *dir, uint32_t dir_n, hipStream_t s);
// ls[i] = {doc_len[i], bits of doc_score[i]}
void launch_pack_len_score(const uint32_t *doc_len, const float *doc_score, uint32_t n, void *ls, hipStream_t s);
struct HybridReduceArgs {
  uint32_t n_tiles, top_n, k;
  uint32_t surv_cap;                   // survivors the reduce workgroup ranks in LDS (<= 4096); more: *out_n = 0xFFFFFFFF
  const uint32_t *tile_hits;
  const uint64_t *part_skey;
  const uint32_t *part_sidx;
  const uint64_t *part_knn;
  // pinned host memory
  uint32_t *out_hits;
  uint64_t *out_skeys;                 // [top_n]
  uint32_t *out_sids, *out_sn;
  uint32_t *out_krows, *out_kkeys, *out_kids, *out_kn;  // [k]
  uint64_t *trace;                     // NULL, or [2][kHybTracePhases] clock readings of the two branches (diagnostics)
  uint32_t *done;                      // pinned host memory, [3] (score branch, KNN branch, hit count): set to 1 -- system-scope
                                       // release -- once that workgroup's answers are in host memory; the host polls them
};
// ---- several callers' queries in ONE grid (round 6: the hybrid coalescer, hybrid_entry.hpp) -----------------------------------
// RediSearch issues queries from a pool of worker threads (src/util/workers.c:58,104); a query's 2 443 tiles are one and a half
// rounds over the chip's ~1 536 resident workgroups and its reduce launch waits for the last of them.  Queries that meet in the
// coalescer's queue share one tile grid (block -> (query, tile): one query's tiles after the other's; knob: dealt out in turn) and
// one reduce launch (three workgroups per query); every query keeps its own FIXED output slots and
// its own pinned answers / completion flags, so the answers are those of its own two launches bit for bit.
// The kernel arguments hold up to kHybBatchMax descriptions side by side (4 KB of kernarg): the flat-AND form reads a tenth of
// ScoreParams -- ScoreParamsFlat carries those fields under the same names (score_one<false, FLAT> is generic in its type).
struct ScoreParamsFlat {
  int scorer, n_groups;
  double avg_doc_len, root_weight, min_score, inv_tanh;
  double idf[kHybMaxLists], bm25_idf[kHybMaxLists], weight[kHybMaxLists];
  int slop;
  long long table_off;
};
struct HybridTileLite {  // HybridTileArgs without the diagnostics; same names, same meaning
  int n;
  const uint32_t *ids[kHybMaxLists];
  const uint32_t *freq[kHybMaxLists];
  uint32_t len[kHybMaxLists];
  long long add[kHybMaxLists];
  uint32_t top_n;
  ScoreParamsFlat P;
  const uint32_t *doc_len;
  const float *doc_score;
  const uint32_t *max_freq;
  uint32_t table_n;
  uint32_t k;
  const void *rows;
  uint32_t stride16, chunks;
  int G, ITERS;
  const void *query;
  uint64_t ids_base;
  LabelRows L;
  uint32_t *tile_hits;
  uint64_t *part_skey;
  uint32_t *part_sidx;
  uint64_t *part_knn;
  uint32_t pool_words;
  const uint32_t *dir[kHybMaxLists];
  uint32_t dir_shift[kHybMaxLists], dir_n[kHybMaxLists];
  const uint2 *len_score;
  int knn_pipeline;
  static constexpr uint64_t *trace = nullptr;
};
constexpr int kHybBatchMax = 7;
struct HybridTileBatch {
  uint32_t n_q;                       // 1 .. kHybBatchMax queries, ascending by their number of tiles
  uint32_t interleave;                // 1: block b of the segment where queries j.. are alive -> tile seg_tile0 + b / alive of query j + b % alive
                                      // 0: the queries' tiles one query after the other
  uint32_t tile_end[kHybBatchMax];    // interleave: blocks before the end of segment j; else blocks before the end of query j
  uint32_t n_tiles[kHybBatchMax];
  HybridTileLite q[kHybBatchMax];
};
static_assert(sizeof(HybridTileBatch) <= 4096, "the batch rides in the kernel arguments");
struct HybridReduceBatch {
  uint32_t n_q;
  HybridReduceArgs q[kHybBatchMax];   // (trace must be NULL)
};
// the description of a query for a shared grid (G / ITERS / pool_words filled as launch_hybrid_tiles fills them); lds_out: the
// dynamic LDS its tiles need
HybridTileLite hybrid_tile_lite(const HybridTileArgs &a, size_t *lds_out);
// false: nothing launched (no instantiation: the caller launches each query on its own)
bool launch_hybrid_tiles_batch(HybridTileBatch &b, int type, int metric, size_t lds, hipStream_t s);
void launch_hybrid_reduce_batch(const HybridReduceBatch &r, hipStream_t s);
uint32_t hybrid_tiles(uint32_t n0);
// type / metric: the index's kernel type and metric (kernels.hpp KT_* / KM_*); false: the staged pipeline takes the query
bool hybrid_tile_supported(int type, int metric, uint32_t stride16, uint32_t n_tiles, uint32_t top_n, uint32_t k);
void launch_hybrid_tiles(const HybridTileArgs &a, int type, int metric, uint32_t n_tiles, hipStream_t s);
void launch_hybrid_reduce(const HybridReduceArgs &r, hipStream_t s);

// ---- the general form of the tile kernel (round 4, hybrid_kernels.hip: hybrid_tree_tile_kernel) ---------------------------
// A root INTERSECTION whose children are terms, unions of terms or intersections of terms (a two-level RSGPU_TreeQuery), with
// max_slop / in_order, per-hit slop for the scorers that divide by it, and -- when the caller wants it -- the ordered hit list
// itself: everything RSGPU_HybridQuery's two-launch form (above) leaves to the staged pipeline except BM25STD.NORM.
//   * up to kHybTreeMaxLists posting lists; list 0 -- a term leaf of the tree, the shortest one -- drives, every other list is
//     probed through its window as above; a driver is a candidate when every REQUIRED set of lists (req[r]: a term, each
//     term of a child intersection, any term of a child union) holds it;
//   * candidates are compacted in DRIVER ORDER (= doc-id order) with their entry index in every leaf's list (0xFFFFFFFF: a
//     union leaf that does not hold the document); max_slop / in_order filters them where they sit compacted
//     (prox_within_range, the code of prox_filter_kernel), a second ordered compaction follows;
//   * per hit: frequencies gathered per LEAF (0 where absent), the slop from the term offsets (prox_min_offset_delta) when
//     the scorer wants it, the score through the two-level fold of score_one (the result's child order: ScoreParams groups);
//   * hit list wanted: doc id | per-leaf frequency | per-leaf entry index at the tile's FIXED slots (tile * 1024 + rank),
//     hybrid_hits_pack_kernel -- a third launch, behind the reduce kernel: the answers do not wait for it -- moves them to
//     their place in the list (exclusive sum of the tiles' hit counts).
constexpr int kHybTreeMaxLists = 8;
constexpr int kHybDeepLevels = 8;  // the deepest result tree (levels below the root) the general tile kernel scores: RSGPU_HybridTreeNodesQuery (4 until round 6)
struct HybridOffsetView {  // OffsetView over the tree's leaves
  const uint8_t *bytes[kHybTreeMaxLists];
  const uint32_t *off_pos[kHybTreeMaxLists];
  const uint32_t *off_len[kHybTreeMaxLists];
};
struct HybridTreeArgs {
  int n;                                       // lists probed: the n_leaves leaves of the result tree, then the excluded (NOT) lists
  int n_leaves;
  uint32_t veto;                               // bit l: a document list l holds is not a hit
  const uint32_t *ids[kHybTreeMaxLists];       // by LIST (probe order: list 0 drives)
  uint32_t len[kHybTreeMaxLists];
  long long add[kHybTreeMaxLists];
  const uint32_t *dir[kHybTreeMaxLists];       // bucket directories (NULL: wave-wide searches)
  uint32_t dir_shift[kHybTreeMaxLists], dir_n[kHybTreeMaxLists];
  uint8_t leaf_of[kHybTreeMaxLists];           // list l -> its leaf column in the result tree (0xFF: an excluded list, no column)
  int n_req;
  uint32_t req[kHybTreeMaxLists];              // bit l: list l; a candidate matches a list of EVERY set
  // round 5 -- a root UNION, one pass per child (the pass's child drives; search_abi.cpp hybrid_general): a document that an
  // EARLIER child matches is that child's pass's hit -- veto_all[v]: every list of the set holds it -> not a hit of this pass;
  // any OTHER child that is an intersection contributes its terms only when it matches as a whole -- opt_all[o]: unless every
  // list of the set holds the document, its leaves count as absent (frequency 0, union_flat.rs:297-320: the result holds the
  // matched children only)
  int n_veto_all, n_opt_all;
  uint32_t veto_all[kHybTreeMaxLists], opt_all[kHybTreeMaxLists];
  const uint32_t *lfreq[kHybTreeMaxLists];     // by LEAF: decoded frequencies (NULL: the codec stores none -- 1)
  // proximity (by leaf): X.max_slop / X.in_order for the filter, the children for both
  int prox_filter, prox_slop;
  ProxParams X;
  HybridOffsetView O;
  // scoring: the two-level result tree in P (groups over the leaf columns); P.slops must be NULL
  uint32_t top_n;
  ScoreParams P;
  const uint32_t *doc_len;
  const float *doc_score;
  const uint32_t *max_freq;
  uint32_t table_n;
  const uint2 *len_score;
  // KNN (as HybridTileArgs)
  uint32_t k;
  const void *rows;
  uint32_t stride16, chunks;
  int G, ITERS;
  const void *query;
  uint64_t ids_base;
  LabelRows L;
  int knn_pipeline;
  // per tile, fixed slots (as HybridTileArgs)
  uint32_t *tile_hits;
  uint64_t *part_skey;
  uint32_t *part_sidx;
  uint64_t *part_knn;
  // the hit list at the tiles' fixed slots (NULL: not wanted)
  uint32_t *hit_ids;                           // [n_tiles * 1024]
  uint32_t *hit_freqs;                         // [n][hit_stride]
  uint32_t *hit_epos;                          // [n][hit_stride], NULL: no list stores offsets
  uint32_t hit_stride;
  uint32_t pool_words;                         // set by the launcher
  // round 6 -- nested trees the sets above cannot express (a union below an intersection below a union: `a (b | (c (d|e)))`): the
  // kernel folds the MATCH over the result tree in P.node_* (post-order, leaves by column): a term matches when its list holds the
  // document, a union when a child does, an intersection when every child does; a node is IN THE RESULT when it matches and its
  // parent is in the result (union_flat.rs:297-320, intersection.rs:256-288) -- a hit = the root matches, the leaves outside the
  // result count as absent.  req / opt_all are not read then; veto still is.
  int tree_pred;
};
bool hybrid_tree_supported(int type, int metric, uint32_t stride16, uint32_t n_tiles, uint32_t top_n, uint32_t k, int n_lists);
void launch_hybrid_tree_tiles(const HybridTreeArgs &a, int type, int metric, uint32_t n_tiles, hipStream_t s);
// out[e] = skey[e] == tau ? sidx[e] : none (settling a tie across the passes of one query)
void launch_hybrid_tie_ids(const uint64_t *skey, const uint32_t *sidx, uint32_t n, uint64_t tau, uint64_t *out, hipStream_t s);
// dst[off(t) + r] = src[t * 1024 + r], r < tile_hits[t], off(t) = sum of tile_hits below t; ids, then n leaf columns of freqs
// (src stride src_stride, dst stride dst_cap) and -- unless NULL -- of entry indices; *total_out (device or pinned) = the sum
// runs (round 6: the hit list of a query of several passes -- a root union, a root of unions): pass p's tiles start at
// runs->first_tile[p]; the workgroup of that tile also writes run_start[p] = off(first_tile[p]), the last one run_start[n] = the sum
struct HybridRuns {
  uint32_t n;                 // passes (<= 8; 0: no runs)
  uint32_t first_tile[9];     // first_tile[n] = n_tiles
  uint32_t *run_start;        // device, [n + 1]
};
void launch_hybrid_hits_pack(const uint32_t *tile_hits, uint32_t n_tiles, int n_leaves, const uint32_t *src_ids,
                             const uint32_t *src_freqs, const uint32_t *src_epos, uint32_t src_stride, uint32_t *dst_ids,
                             uint32_t *dst_freqs, uint32_t *dst_epos, uint32_t dst_cap, uint32_t *total_out, hipStream_t s,
                             const HybridRuns *runs = nullptr);
// The packed runs -- each ascending by doc id, no doc id in two of them (every hit is reported by exactly one pass) -- merged into
// ONE ascending list (union_flat.rs:223-320 yields a union's documents in doc-id order): entry j of run p goes to its index in the
// run + the number of entries below its doc id in every other run (a binary search per other run).  src / dst as the pack's dst.
void launch_hybrid_hits_merge(const HybridRuns &runs, int n_leaves, const uint32_t *src_ids, const uint32_t *src_freqs,
                              const uint32_t *src_epos, uint32_t src_cap, uint32_t *dst_ids, uint32_t *dst_freqs, uint32_t *dst_epos,
                              uint32_t dst_cap, uint32_t max_total, hipStream_t s);

// ---- FT.HYBRID fusion (fusion_kernels.hip) -------------------------------------------------------------
constexpr uint32_t kFuseMaxWindow = 4096;  // per upstream: (2 * 4096) * 17 bytes of LDS
struct FuseParams {
  int scoring;  // 0 RRF, 1 LINEAR
  double constant, w0, w1;
  int metric;  // VecSimMetric of the vector upstream's distances, < 0: b_scores are final scores
  const uint64_t *a_ids, *b_ids;      // device, already cut to the window
  const double *a_scores, *b_scores;
  uint32_t na, nb;
  uint64_t *ids_out;   // [na + nb]
  double *scores_out;  // [na + nb]
  uint32_t *count_out;
};
void launch_hybrid_fuse(const FuseParams &p, hipStream_t s);

}  // namespace rsgpu
