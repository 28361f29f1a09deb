/*
 * rsgpu_search.h -- C ABI of the integer / scoring half of the hot path on MI355X: posting-list
 * decode, N-way intersection, the built-in scorers and the score top-N, plus the hybrid
 * (pre-filter -> FLAT ad-hoc KNN) step.
 *
 * What each entry point replaces in the reference:
 *   RSGPU_Postings_Upload / _Decode   the block reader + codecs
 *                                     (src/redisearch_rs/inverted_index/src/reader/core.rs,
 *                                      codec/{full,freqs_only,doc_ids_only,...}.rs, qint/src/lib.rs:139-214)
 *   RSGPU_Intersect                   Intersection::read / find_consensus
 *                                     (src/redisearch_rs/rqe_iterators/src/intersection.rs:256-288,428-452)
 *                                     C constructor NewIntersectionIterator, headers/iterators_ffi.h:309
 *   RSGPU_Hits_Score                  the RSScoringFunction loop rpscoreNext drives
 *                                     (src/result_processor.c:570-603 over src/ext/default.c:68-461)
 *   RSGPU_Hits_TopN                   rpsortNext_innerLoop + cmpByScore (src/result_processor.c:752-850)
 *   RSGPU_Hits_KnnRerank              HybridIterator ADHOC_BF: computeDistances_RAM
 *                                     (src/iterators/hybrid_reader.c:289-335)
 *   RSGPU_CalculateIDF / _BM25        CalculateIDF / CalculateIDF_BM25
 *                                     (src/redisearch_rs/c_entrypoint/idf_ffi/src/lib.rs -> idf/src/lib.rs:67-108)
 *
 * These are batch-granular on purpose: the reference's per-result virtual calls (one Read(), one
 * scorer call per hit) cannot feed a GPU (SURVEY.md 8b boundary 2/3).  INTEGRATION.md shows the
 * adapter that hands a hit list back to the C pipeline through NewSortedIdListIterator /
 * NewMetricIteratorSortedByScore (headers/iterators_ffi.h:574,480).
 *
 * The posting bytes are uploaded AS-IS in the reference's block format; decoding happens on the
 * device.  Doc ids are 64-bit (t_docId) at this interface.  On the device a list holds 32-bit offsets from its own
 * base -- 0 while every id of the list fits 32 bits, else the list's first doc id -- and the lists of one query are
 * re-based on the fly onto the smallest doc id among them: a list, and the lists combined in one query, must span fewer
 * than 2^32 doc ids (ids are assigned monotonically, so the live window of an index does).  Term offsets stay where they
 * are -- in the encoded bytes -- and are read in place by the proximity kernels (max_slop / in_order, the scorers' slop).
 */
#ifndef RSGPU_SEARCH_H
#define RSGPU_SEARCH_H

#include <stddef.h>
#include <stdint.h>

#include "VecSim/vec_sim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* posting codecs (record layouts under inverted_index/src/codec/) */
typedef enum {
  RSGPU_CODEC_FULL = 0,           /* qint[delta,freq,fieldMask,offsetsLen] + offsets bytes */
  RSGPU_CODEC_FREQS_FIELDS = 1,   /* qint[delta,freq,fieldMask] */
  RSGPU_CODEC_FREQS_ONLY = 2,     /* qint[delta,freq] */
  RSGPU_CODEC_FIELDS_ONLY = 3,    /* qint[delta,fieldMask] */
  RSGPU_CODEC_FIELDS_OFFSETS = 4, /* qint[delta,fieldMask,offsetsLen] + offsets */
  RSGPU_CODEC_OFFSETS_ONLY = 5,   /* qint[delta,offsetsLen] + offsets */
  RSGPU_CODEC_FREQS_OFFSETS = 6,  /* qint[delta,freq,offsetsLen] + offsets */
  RSGPU_CODEC_DOCIDS_ONLY = 7,    /* varint delta */
  RSGPU_CODEC_RAW_DOCIDS = 8,     /* u32 delta from the block's first doc id */
  /* the *Wide variants for schemas with more than 32 text fields: the field mask (up to 128 bits) is a varint
   * behind the qint fields (reference codec/full.rs:195, freqs_fields.rs:112, fields_only.rs:107, fields_offsets.rs:137) */
  RSGPU_CODEC_FULL_WIDE = 9,           /* qint[delta,freq,offsetsLen] + varint mask + offsets */
  RSGPU_CODEC_FREQS_FIELDS_WIDE = 10,  /* qint[delta,freq] + varint mask */
  RSGPU_CODEC_FIELDS_ONLY_WIDE = 11,   /* varint delta + varint mask */
  RSGPU_CODEC_FIELDS_OFFSETS_WIDE = 12 /* qint[delta,offsetsLen] + varint mask + offsets */
} RSGPU_Codec;

/* scorers registered by DefaultExtensionInit (reference src/ext/default.c:737-) */
typedef enum {
  RSGPU_SCORER_BM25STD = 0,
  RSGPU_SCORER_BM25STD_TANH = 1,
  RSGPU_SCORER_BM25 = 2,
  RSGPU_SCORER_TFIDF = 3,
  RSGPU_SCORER_TFIDF_DOCNORM = 4,
  RSGPU_SCORER_DOCSCORE = 5,
  RSGPU_SCORER_DISMAX = 6,
  /* BM25STD followed by the max-score normaliser the reference chains behind it for `SCORER BM25STD.NORM`
   * (RPMaxScoreNormalizer, src/result_processor.c:1770-1812; pipeline_construction.c:546-547) */
  RSGPU_SCORER_BM25STD_NORM = 7
} RSGPU_Scorer;

typedef struct RSGPU_Postings RSGPU_Postings;
typedef struct RSGPU_Hits RSGPU_Hits;
typedef struct RSGPU_DocTable RSGPU_DocTable;

/* Upload one posting list: per-block headers (IndexBlock first/last doc id, entry count; byte_offset
 * has n_blocks+1 entries into `bytes`). NULL on failure (RSGPU_LastError). */
RSGPU_Postings *RSGPU_Postings_Upload(int codec, size_t n_blocks, const uint64_t *first_doc_id,
                                      const uint64_t *last_doc_id, const uint32_t *num_entries,
                                      const uint64_t *byte_offset, const uint8_t *bytes);
void RSGPU_Postings_Free(RSGPU_Postings *p);
size_t RSGPU_Postings_NumEntries(const RSGPU_Postings *p);
size_t RSGPU_Postings_NumBytes(const RSGPU_Postings *p);
/* Decode every record on the device; any host output may be NULL. Returns #records or -1.  A codec that stores no
 * frequency yields 1 for every record -- the term record's default the reference's reader leaves in place
 * (index_result/src/core/mod.rs:192-197) -- and that is also what such a list contributes to a hit's frequencies. */
long RSGPU_Postings_Decode(RSGPU_Postings *p, uint64_t *doc_ids_out, uint32_t *freqs_out, uint32_t *masks_out);

/* Wide codecs: the 128-bit field mask of every record (low / high 64 bits; either may be NULL; masks_out of
 * RSGPU_Postings_Decode carries the low 32 bits).  Returns #records or -1. */
long RSGPU_Postings_DecodeWideMasks(RSGPU_Postings *p, uint64_t *masks_lo_out, uint64_t *masks_hi_out);

/* Docs present in ALL lists (decode + intersect on the device). Hits are ascending by doc id and carry
 * the matched frequency of every input list. 1..32 lists (the reference's own tests go to 25 children). NULL on
 * failure. */
RSGPU_Hits *RSGPU_Intersect(RSGPU_Postings *const *lists, size_t n_lists);
/* The same with the proximity constraints of NewIntersectionIterator(its, num, max_slop, in_order, weight)
 * (reference headers/iterators_ffi.h:309; Intersection::new_with_slop_order, rqe_iterators/src/intersection.rs:94-119;
 * the check is RSIndexResult::is_within_range, index_result/src/core/proximity.rs:262-298): a document in all lists
 * is kept only if the terms' positions (decoded from the lists' offset bytes on the device) fit a window with at most
 * max_slop foreign tokens -- max_slop < 0: no slop constraint -- and, with in_order, appear in the order of `lists`
 * (which are then NOT re-ordered by size).  Lists whose codec stores no offsets do not take part in the check.
 * The hit list borrows the posting lists (slop-aware scoring reads their offset bytes): free the hits first. */
RSGPU_Hits *RSGPU_IntersectEx(RSGPU_Postings *const *lists, size_t n_lists, long max_slop, int in_order);
/* (round 4: with a window over at most eight lists that store offsets the general hybrid tile kernel builds the list -- probe,
 * window test and ordered write in one launch + a pack launch; RSGPU_HybridQueryPath reads 2 after such a call.  Same hit list.) */
void RSGPU_Hits_Free(RSGPU_Hits *h);
size_t RSGPU_Hits_Len(const RSGPU_Hits *h);
/* doc_ids[len]; freqs[n_lists][len] in the order the lists were given. Either may be NULL. */
int RSGPU_Hits_Read(const RSGPU_Hits *h, uint64_t *doc_ids, uint32_t *freqs);

/* ---- record access: what the iterator seam (include/rs_iterator.h) needs to rebuild, per hit, the RSIndexResult the
 * reference's own iterators would hold in `current` ------------------------------------------------------------------ */
/* Number of term columns ("leaves") of the hit list and, for child slot s of the aggregate, the index of its list in the
 * caller's array: an intersection iterates its children by ascending size unless in_order (reference
 * rqe_iterators/src/intersection.rs:94-119), and the result's children come in THAT order.  Returns #leaves or -1. */
int RSGPU_Postings_Codec(const RSGPU_Postings *p); /* RSGPU_Codec of the list, -1 for NULL */
size_t RSGPU_Hits_NumLeaves(const RSGPU_Hits *h);
int RSGPU_Hits_IsUnion(const RSGPU_Hits *h); /* 1: built by RSGPU_Union (a child may be absent from a hit) */
int RSGPU_Hits_LeafOrder(const RSGPU_Hits *h, int *list_of_child);
/* Shape of the result tree behind the hits: the root (union or intersection) has n_groups children, child g being leaf
 * group_first[g] alone (group_op 0) or a union (1) / intersection (2) of the leaves [group_first[g], group_first[g+1])
 * with weight group_weight[g]; leaves are the child slots of RSGPU_Hits_LeafOrder.  group_first has n_groups + 1
 * entries.  Any output may be NULL.  Returns n_groups or -1. */
int RSGPU_Hits_Tree(const RSGPU_Hits *h, int *root_is_union, int *group_first, int *group_op, double *group_weight);
/* doc ids of hits [first, first+count) (clamped to the hit list); returns #written or -1. */
long RSGPU_Hits_ReadRange(const RSGPU_Hits *h, size_t first, size_t count, uint64_t *doc_ids);
/* For hits [first, first+count) and one list (index in the caller's array): the record the term's reader would have
 * yielded -- entry index in the posting list (0xFFFFFFFF: the list does not hold the document, i.e. a union child that
 * did not match), frequency, field mask (low / high 64 bits), and where the record's term-offsets blob lies in the
 * uploaded bytes (position, length; RSGPU_Postings_ReadBytes fetches them).  Any output may be NULL.  The lists must
 * still be alive.  Returns #written or -1. */
long RSGPU_Hits_ReadRecords(const RSGPU_Hits *h, size_t list, size_t first, size_t count, uint32_t *entry, uint32_t *freqs,
                            uint64_t *mask_lo, uint64_t *mask_hi, uint64_t *offsets_pos, uint32_t *offsets_len);
/* bytes [pos, pos+len) of the list as uploaded (device -> host).  0 on success. */
int RSGPU_Postings_ReadBytes(const RSGPU_Postings *p, size_t pos, size_t len, uint8_t *out);

/* Per-document scorer inputs (RSDocumentMetadata: score, maxTermFreq, docLen -- reference
 * src/redisearch.h:97-132), arrays indexed by doc id, n = max doc id + 1. */
RSGPU_DocTable *RSGPU_DocTable_Upload(size_t n, const uint32_t *doc_len, const float *doc_score,
                                      const uint32_t *max_term_freq);
/* The same for a window of doc ids: entry j describes doc id first_doc_id + j (indexes whose ids have grown past 2^32
 * keep a table of the live window only).  Hits outside the window score as unknown documents. */
RSGPU_DocTable *RSGPU_DocTable_UploadWindow(uint64_t first_doc_id, size_t n, const uint32_t *doc_len,
                                            const float *doc_score, const uint32_t *max_term_freq);
void RSGPU_DocTable_Free(RSGPU_DocTable *t);

typedef struct {
  int scorer;           /* RSGPU_Scorer */
  size_t num_docs;      /* ScoringFunctionArgs.indexStats.numDocs */
  double avg_doc_len;   /* indexStats.avgDocLen */
  uint64_t tanh_factor; /* BM25STD.TANH */
  double root_weight;   /* weight of the intersection node */
  double min_score;     /* early-out threshold of TFIDF / legacy BM25 (0 = off) */
  const double *idf;      /* [n_lists] QueryTerm_GetIDF */
  const double *bm25_idf; /* [n_lists] QueryTerm_GetBM25_IDF */
  const double *weight;   /* [n_lists] term node weights */
} RSGPU_ScoreArgs;

/* Score every hit as Intersection{Term...} (fp64, the C source's float constants). scores_out (host,
 * [len]) may be NULL; the scores also stay on the device for _TopN. 0 on success. */
int RSGPU_Hits_Score(RSGPU_Hits *h, const RSGPU_DocTable *t, const RSGPU_ScoreArgs *args, double *scores_out);
/* n best hits by (score descending, doc id ascending). Returns #written or -1. */
long RSGPU_Hits_TopN(RSGPU_Hits *h, size_t n, uint64_t *doc_ids_out, double *scores_out);
/* Ad-hoc brute force over the hits: k nearest (distance ascending, doc id ascending) among the hits
 * that exist in the FLAT index. Returns #written or -1. */
long RSGPU_Hits_KnnRerank(RSGPU_Hits *h, VecSimIndex *index, const void *query, size_t k, uint64_t *doc_ids_out,
                          double *dist_out);

/* The whole hybrid query in ONE call (BASELINE configs[4]): intersection of `lists`, then -- both optional, run
 * concurrently on two streams -- (a) score every hit + the top_n by (score desc, doc id asc), (b) the k nearest hits of
 * a FLAT index (ad-hoc brute force).  Two stream synchronisations in total, against one per stage (and several inside
 * the selections) for the stage-by-stage entry points; results are identical to calling
 * RSGPU_Intersect / _Hits_Score / _Hits_TopN / _Hits_KnnRerank in sequence.  0 on success. */
typedef struct {
  RSGPU_Postings *const *lists; /* in: 1..32 intersection children */
  size_t n_lists;
  const RSGPU_DocTable *table;  /* in: (a) needs table, score and top_n > 0 */
  const RSGPU_ScoreArgs *score;
  size_t top_n;
  VecSimIndex *index;           /* in: (b) needs index, query and k > 0 */
  const void *query;
  size_t k;
  uint64_t *top_ids;            /* out [top_n] */
  double *top_scores;           /* out [top_n] */
  uint64_t *knn_ids;            /* out [k] */
  double *knn_dists;            /* out [k] */
  size_t n_hits, n_top, n_knn;  /* out: intersection size, entries written to top_* / knn_* */
  RSGPU_Hits **hits_out;        /* optional out: the hit list itself (caller frees); NULL = dropped */
  /* round 6 -- the query's deadline.  The reference polls TimedOut_WithCtx per candidate (src/iterators/hybrid_reader.c:311,
   * src/util/timeout.h:57-100), its iterators return ITERATOR_TIMEOUT (src/iterators/iterator_api.h), and VecSim polls
   * timeoutCallback(queryParams->timeoutCtx) (src/module-init/module-init.c:150): the same callback shape.  NULL = no deadline.
   * Polled on entry (at least once, however small the query), while the host waits for the device, and between the stages of
   * the staged forms; when it returns non-zero the call waits for what it has in flight, returns RSGPU_TIMED_OUT with n_hits =
   * n_top = n_knn = 0 and no hit list, and may be issued again.  (A caller that zero-initialises the block gets no deadline.) */
  int (*timeout_cb)(void *ctx);
  void *timeout_ctx;
} RSGPU_HybridQueryArgs;
#define RSGPU_TIMED_OUT 1       /* RSGPU_HybridQuery / _TreeQuery / _TreeNodesQuery: 0 ok, -1 error (RSGPU_LastError), 1 deadline */
int RSGPU_HybridQuery(RSGPU_HybridQueryArgs *args);
/* how the calling thread's last RSGPU_HybridQuery / RSGPU_HybridTreeQuery ran: 0 = the staged pipeline (intersection written
 * out, score / top-N and KNN branches on two streams; stage by stage for trees), 1 = two launches (no hits_out, a flat AND of
 * <= 4 term lists, top_n / k <= 64 (63 for BM25STD.NORM), no slop-dependent scorer over lists with offsets: one tile kernel -- probe, scores,
 * distances, per-tile winners -- and one reduce kernel), 2 = the general tile kernel + the reduce kernel (<= 8 lists under a
 * root intersection of terms / unions of terms / intersections of terms, max_slop / in_order, per-hit slop from the term
 * offsets, NOT children, BM25STD.NORM; hits_out wanted: a third launch packs the list; round 5: a root union of terms /
 * intersections of terms -- one pass per child --, a root intersection whose children are all unions -- the smallest union
 * drives, one pass per term of it -- and, through RSGPU_HybridTreeNodesQuery, nested trees; round 6: hits_out of those
 * several-pass queries too -- every pass reports its hits in doc-id order, a document by exactly one pass, and one more launch
 * merges the packed runs by doc id: union_flat.rs:223-320 yields a union's documents in doc-id order).
 * top_n / k > 64 and indexes whose labels no device table holds stay staged.  Same answers. */
int RSGPU_HybridQueryPath(void);
/* The hybrid coalescer (round 6; knobs "hybrid_coalesce" 1, "hybrid_coalesce_depth" 2, "hybrid_coalesce_interleave" 0 of
 * RSGPU_SetTuning): RediSearch issues queries from a pool of worker threads (src/util/workers.c:58,104; the hybrid iterator's
 * loop, src/iterators/hybrid_reader.c:309-327, runs on each).  Two-launch queries (path 1) of concurrent callers share tile grids: at
 * most `depth` grids are in flight per device; a caller that arrives below that launches at once, one that arrives at it queues and is
 * launched -- with up to six others, as ONE grid + ONE reduce launch -- by the caller whose grid finishes next.  Every query keeps
 * its own output slots and pinned answers: replies are bit-identical to serial ones; a single caller never queues.
 * out[0] queries launched alone, [1] shared grids, [2] queries they carried, [3] queries that queued, [4] queries that re-launched
 * alone after a shared launch failed.  reset != 0: the counters go back to zero after they were read. */
void RSGPU_GetHybridCoalesceStats(uint64_t out[5], int reset);
/* diagnostics (RSGPU_SetTuning("hybrid_trace", 1)): the phase clock of every tile of the calling thread's last two-launch query,
 * out[tile * 9 + phase] readings of the 100 MHz device clock; returns the number of tiles copied (0: no trace), -1 on error */
long RSGPU_HybridTrace(uint64_t *out, size_t cap_tiles);

/* Union of 1..32 lists: documents present in ANY list, ascending doc id; a list that does not hold the document
 * contributes freq 0 (reference rqe_iterators/src/union_flat.rs:223-257,297-320).  Scoring a union hit list
 * follows the reference's Union node: absent children add nothing, the slop divisor counts the matched
 * children only, DISMAX takes the children's maximum.  Returns NULL on error. */
RSGPU_Hits *RSGPU_Union(RSGPU_Postings *const *lists, size_t n_lists);
/* Two-level query tree: the root (AND / OR) over `n_groups` children, child g being the single list group_first[g]
 * (RSGPU_OP_TERM) or the OR / AND of lists[group_first[g] .. group_first[g+1]) with weight group_weight[g] -- e.g. the
 * stemmer's (run|running|ran) (shoe|shoes), or (a b) | (c d).  The hit list carries every term's frequency (0 where a
 * union child did not match), so RSGPU_Hits_Score evaluates the reference's result tree Intersection{Union{..},..}
 * (src/ext/default.c recursions: an aggregate sums its children and multiplies by its weight, DISMAX takes a union's
 * maximum) and its slop merges a union child's term positions (proximity.rs OffsetIter::Merge).  max_slop / in_order
 * apply to a root intersection.  idf / bm25_idf / weight of RSGPU_ScoreArgs are per LIST, in the order of `lists`. */
#define RSGPU_OP_TERM 0
#define RSGPU_OP_UNION 1
#define RSGPU_OP_INTERSECT 2
/* The tree is evaluated AS GIVEN: an aggregate with a single child stays an aggregate (its weight applies, the result holds
 * an Intersection / Union record around the child).  The reference reduces such nodes BEFORE it builds iterators -- an
 * intersection with one real child becomes that child, wildcards are stripped, an empty child empties the whole
 * (new_intersection_iterator, rqe_iterators/src/intersection.rs:354-420; the union reducer likewise) -- so a caller that
 * mirrors the query AST applies those reductions first (RSGPU_EvalTreeNodes: the same). */
typedef struct {
  int root_op;                 /* RSGPU_OP_INTERSECT or RSGPU_OP_UNION */
  size_t n_groups;
  const size_t *group_first;   /* [n_groups + 1] */
  const int *group_op;         /* [n_groups], NULL: all RSGPU_OP_TERM */
  const double *group_weight;  /* [n_groups], NULL: all 1.0 (a term's own weight stays in RSGPU_ScoreArgs.weight) */
  RSGPU_Postings *const *lists;
  long max_slop;               /* < 0: none */
  int in_order;
} RSGPU_TreeQuery;
RSGPU_Hits *RSGPU_EvalTree(const RSGPU_TreeQuery *q);
/* (round 4: a root intersection with an aggregate child over at most eight lists, one of them a term every hit must hold, is
 * evaluated by the general hybrid tile kernel -- every list probed in place, no child hit list built first; the calling
 * thread's RSGPU_HybridQueryPath reads 2 after such a call, 0 after a stage-by-stage one.  Same hit list either way.) */
/* RSGPU_HybridQuery over such a tree -- the filter the reference hands its hybrid iterator as `childIt`
 * (src/iterators/hybrid_reader.c:625 NewHybridVectorIterator; intersections with max_slop / in_order:
 * rqe_iterators/src/intersection.rs:94-119) -- in one call: args->lists / n_lists are ignored (tree->lists are the terms;
 * RSGPU_ScoreArgs.idf / bm25_idf / weight per LIST in their order), everything else as RSGPU_HybridQuery, hits_out included.
 * Results are those of RSGPU_EvalTree + RSGPU_Hits_Score / _TopN / _KnnRerank; RSGPU_HybridQueryPath tells how it ran. */
int RSGPU_HybridTreeQuery(const RSGPU_TreeQuery *tree, RSGPU_HybridQueryArgs *args);
/* For RSGPU_HybridTreeQuery only: a child of a root INTERSECTION that EXCLUDES documents -- `a -b`, `a (b|c) -(d|e)`: the group's
 * lists are the excluded terms (a document that any of them holds is not a hit).  The reference's Not iterator yields a VIRTUAL
 * result of frequency 0 (rqe_iterators/src/not.rs:106-118, index_result/src/core/mod.rs:103-112): it adds nothing to any scorer's
 * sum and has no offsets, but it IS a child of the intersection's result -- IndexResult_MinOffsetDelta counts it (the offset-less
 * slop is children - 1).  Its lists take no idf / weight (leave their RSGPU_ScoreArgs entries 0) and have no column in the hit
 * list: hits_out (round 5) receives the positive children's columns and the result tree with the virtual child in it --
 * RSGPU_Hits_Score on that list gives the scores the query ranked by.  The query runs on the general tile kernel or not at all
 * (more than eight lists, no term to drive, a general label map: -1 with a message; RSGPU_EvalTree rejects the operator). */
#define RSGPU_OP_NOT 3

/* Query trees of ANY depth: `nodes` in POST-ORDER -- a term names its list; an aggregate (RSGPU_OP_UNION /
 * RSGPU_OP_INTERSECT) takes the n_children complete subtrees immediately before it; the last node is the root (its weight
 * is RSGPU_ScoreArgs.root_weight).  Every intersection node may carry its own max_slop (< 0: none) / in_order, as every
 * NewIntersectionIterator does (reference headers/iterators_ffi.h:309).  Each list appears at most once; at most 32
 * terms, 64 nodes, 16 levels.  The hit list carries every term's frequency and the whole result tree:
 * RSGPU_Hits_Score evaluates it as the reference's recursions do (src/ext/default.c:68-106,164-209,253-302,378-455 --
 * an aggregate sums its children in the result's child order and multiplies by its weight, DISMAX takes a union's
 * maximum), for any nesting.  RSGPU_ScoreArgs.idf / bm25_idf / weight are per LIST, in the order of `lists`. */
typedef struct {
  int op;             /* RSGPU_OP_TERM / _UNION / _INTERSECT */
  size_t list;        /* term: index into `lists` */
  size_t n_children;  /* aggregate */
  double weight;      /* aggregate: the node's weight (a term's stays in RSGPU_ScoreArgs.weight) */
  long max_slop;      /* intersection: < 0 none */
  int in_order;       /* intersection */
} RSGPU_TreeNode;
RSGPU_Hits *RSGPU_EvalTreeNodes(const RSGPU_TreeNode *nodes, size_t n_nodes, RSGPU_Postings *const *lists, size_t n_lists);
/* RSGPU_HybridQuery over such a tree (round 5) -- the filter the reference hands its hybrid iterator may nest to any depth
 * (src/iterators/hybrid_reader.c:625; `a ((b c)|d)`, `a (b|(c d)) (e|f)`): args->lists / n_lists are the terms the nodes name,
 * RSGPU_ScoreArgs.idf / bm25_idf / weight per LIST in their order, everything else as RSGPU_HybridQuery, hits_out included.
 * Results are those of RSGPU_EvalTreeNodes + RSGPU_Hits_Score / _TopN / _KnnRerank.  RSGPU_HybridQueryPath reads 2 when the
 * general tile kernel took it: a root intersection over <= 8 lists, one of them a term every hit holds (or a child that is a plain
 * union of terms to drive it), nested at most eight levels (four until round 6), unions and intersections in any arrangement (round
 * 6: a union below an intersection below a union too -- the kernel folds the match over the result tree), no max_slop / in_order
 * BELOW the root (on the root: yes, also over nested children -- round 6); hits_out (round 6) and slop-dependent scorers over lists that
 * store offsets under nested children included (round 6: a child's offsets are its leaves' in the result, merged);
 * 0 when it ran stage by stage.  RSGPU_OP_NOT nodes (children: the excluded terms) are accepted as children of the root
 * intersection -- `a ((b c)|d) -e` -- with the meaning they have in RSGPU_HybridTreeQuery; such a query has no staged form (a shape
 * the tile kernel declines is refused: -1, RSGPU_LastError says so); hits_out (round 6) receives the positive children's columns and
 * the result tree with the virtual child in it, as RSGPU_HybridTreeQuery's. */
int RSGPU_HybridTreeNodesQuery(const RSGPU_TreeNode *nodes, size_t n_nodes, RSGPU_HybridQueryArgs *args);
/* The result tree behind a hit list, post-order (after the intersections sorted their children by size): per node the
 * operator, the leaf column of a term (-1 for aggregates; leaves are the child slots of RSGPU_Hits_LeafOrder), the number
 * of children and the weight.  Arrays of 64 entries suffice; any may be NULL.  Returns the number of nodes or -1. */
int RSGPU_Hits_TreeNodes(const RSGPU_Hits *h, int *op, int *leaf, int *n_children, double *weight);

/* NOT: doc ids 1..max_doc_id the child does not hold (rqe_iterators/src/not.rs:171-209), or -- with a
 * `universe` list of existing documents -- the universe's entries <= max_doc_id the child does not hold
 * (not_optimized.rs).  The hits are virtual results: one child with freq 1; score them with idf = 1
 * (reference src/ext/default.c:289-293).  Returns NULL on error. */
RSGPU_Hits *RSGPU_Not(RSGPU_Postings *child, RSGPU_Postings *universe, uint64_t max_doc_id);

/* FT.HYBRID fusion of the ranked search list (doc ids + scores, best first) and the ranked vector list (doc ids
 * + distances, nearest first): RPHybridMerger + HybridRRFScore / HybridLinearScore
 * (reference src/result_processor.c:2549-2571,2613-2670; src/hybrid/hybrid_scoring.c:41-84) with the vector
 * score normalised by VectorNorm_<metric> (src/vector_normalization.h:37-60; metric < 0: vec_scores are used
 * as given).  scoring 0 = RRF: sum of 1/(constant + rank), rank = 1-based position within its list; 1 = LINEAR:
 * weights[0]*search_score + weights[1]*norm(distance).  At most `window` (<= 4096) entries of each list are
 * consumed.  Writes the fused list (score descending, lower doc id first) cut to top_n; returns #written or -1. */
#define RSGPU_HYBRID_RRF 0
#define RSGPU_HYBRID_LINEAR 1
long RSGPU_HybridFuse(int scoring, double rrf_constant, const double *weights, int metric, const uint64_t *search_ids,
                      const double *search_scores, size_t n_search, const uint64_t *vec_ids, const double *vec_scores,
                      size_t n_vec, size_t window, size_t top_n, uint64_t *doc_ids_out, double *scores_out);

/* idf = logb(1 + (N+1)/max(n,1)); bm25 idf = ln(1 + (max(N,n) - n + 0.5)/(n + 0.5)) */
double RSGPU_CalculateIDF(size_t total_docs, size_t term_docs);
double RSGPU_CalculateIDF_BM25(size_t total_docs, size_t term_docs);

/* GPU milliseconds (HIP events) of the last call of each stage on the calling thread */
void RSGPU_SearchProfile(double *decode_ms, double *intersect_ms, double *score_ms, double *topn_ms, double *knn_ms);

#ifdef __cplusplus
}
#endif
#endif /* RSGPU_SEARCH_H */
