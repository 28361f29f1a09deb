/*
 * rs_iterator.h -- Boundary 3 of SURVEY.md 8(b): the reference's query-iterator vtable, served from MI355X hit lists.
 *
 * RediSearch evaluates a query as a tree of `QueryIterator`s (reference src/iterators/iterator_api.h:46-151): the
 * result processor pulls one document at a time through Read(), composite iterators steer their children with
 * SkipTo(), and `current` holds the RSIndexResult the scorers walk.  The constructors the C pipeline calls are
 * declared in src/redisearch_rs/headers/iterators_ffi.h -- NewIntersectionIterator(its, num, max_slop, in_order,
 * weight) :309, NewUnionIterator, NewNotIterator, NewInvIndIterator_TermQuery(idx, sctx, fieldMaskOrIndex, term,
 * weight) :404.
 *
 * The iterators below have the same struct, the same status codes and the same observable behaviour
 * (rqe_iterators/src/intersection.rs:428-530, union_flat.rs, not.rs) but take posting lists that live on the GPU
 * (include/rsgpu_search.h) instead of child iterators: the whole AND / OR / NOT is evaluated on the device when the
 * iterator is created, and Read / SkipTo / Rewind walk the resulting hit list, paging doc ids and per-term records
 * (frequency, field mask, term offsets) back in blocks.  `current` is the tree the reference's own iterators hold:
 * an Intersection / Union aggregate over Term records in the children's iteration order (ascending size unless
 * in_order), or a Virtual result for NOT -- so NewHybridVectorIterator takes one as its `childIt`
 * (src/vector_index.c:262-290) and rpscoreNext scores its results (src/result_processor.c:570-603) unchanged.
 *
 * The RSIndexResult constructors are Rust inside the module (src/redisearch_rs/headers/types_ffi.h:89,233,263,331,
 * 352,358,364,444); this library reaches them through a table (RSGPU_ResultAPI) that is filled from the process's
 * symbols at first use, or set explicitly.  When the file is compiled INTO the module the table is bound at link time.
 *
 * The struct layouts are shared memory between module and library: tests/test_iterator_abi.py compiles a probe against
 * the reference's own headers (where /root/reference exists) and compares every sizeof / offsetof / enum value.
 */
#ifndef RSGPU_RS_ITERATOR_H
#define RSGPU_RS_ITERATOR_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "rs_extension.h" /* t_docId, t_fieldMask, RSIndexResult, RSQueryTerm */
#include "rsgpu_search.h" /* RSGPU_Postings, RSGPU_Hits */

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/redisearch_rs/headers/rqe_iterator_type.h:41-80 (the members this library reports) */
enum IteratorType {
  IteratorType_Union = 6,
  IteratorType_Intersect = 7,
  IteratorType_Not = 8,
  IteratorType_Empty = 13,
  IteratorType_IdListSorted = 14,
  IteratorType_Max = 24,
  IteratorType_Force32 = 0x7fffffff
};

/* reference src/iterators/iterator_api.h:23-28 */
typedef enum IteratorStatus {
  ITERATOR_OK,
  ITERATOR_NOTFOUND,
  ITERATOR_EOF,
  ITERATOR_TIMEOUT,
} IteratorStatus;

/* reference src/iterators/iterator_api.h:30-41 */
typedef enum ValidateStatus {
  VALIDATE_OK,
  VALIDATE_MOVED,
  VALIDATE_ABORTED,
  VALIDATE_TIMEOUT,
} ValidateStatus;

struct IndexSpec;
typedef struct MapBuilder RsMapBuilder;
typedef struct ProfilePrintCtx RsProfilePrintCtx;

/* reference src/iterators/iterator_api.h:46-151 -- member for member */
typedef struct QueryIterator {
  enum IteratorType type;
  bool atEOF;            /* set once a Read / SkipTo has returned ITERATOR_EOF, never while positioned on the last result */
  t_docId lastDocId;     /* the last doc id read; 0 before the first read */
  RSIndexResult *current; /* non-NULL after OK / NOTFOUND; NULL before the first read and after EOF */
  size_t (*NumEstimated)(const struct QueryIterator *self);
  IteratorStatus (*Read)(struct QueryIterator *self);
  IteratorStatus (*SkipTo)(struct QueryIterator *self, t_docId docId);
  ValidateStatus (*Revalidate)(struct QueryIterator *self, struct IndexSpec *spec);
  void (*Free)(struct QueryIterator *self);
  void (*Rewind)(struct QueryIterator *self);
  struct QueryIterator *(*ProfileChildren)(struct QueryIterator *self);
  void (*PrintProfile)(const struct QueryIterator *self, RsMapBuilder *map, RsProfilePrintCtx *ctx);
} QueryIterator;

#define RS_FIELDMASK_ALL (~(t_fieldMask)0) /* reference src/redisearch.h */

/* ---- the module's RSIndexResult constructors (reference src/redisearch_rs/headers/types_ffi.h) -------------------- */
typedef struct RSGPU_ResultAPI {
  RSIndexResult *(*NewIntersectResult)(size_t cap, double weight);              /* :331 (children are borrowed) */
  RSIndexResult *(*NewUnionResult)(size_t cap, double weight);                  /* :358 */
  RSIndexResult *(*NewVirtualResult)(double weight, t_fieldMask field_mask);    /* :364 */
  RSIndexResult *(*NewTokenRecord)(RSQueryTerm *term, double weight);           /* :352 (takes the term) */
  void (*AggregateResult_AddChild)(RSIndexResult *parent, RSIndexResult *child); /* :89 */
  void (*IndexResult_AggregateReset)(RSIndexResult *result);                    /* :233 */
  void (*IndexResult_Free)(RSIndexResult *result);                              /* :263 */
  /* RSOffsetVector_SetData(RSOffsetSlice *offsets, const char *data, uint32_t len) :444 -- the slice is the
   * `offsets` member of a Term record (borrowed bytes; they stay valid until the next Read / SkipTo / Rewind) */
  void (*RSOffsetVector_SetData)(void *offsets, const char *data, uint32_t len);
} RSGPU_ResultAPI;
/* Install the table (copied).  NULL: look every name up in `dl_handle` (a dlopen handle; NULL = the whole process,
 * RTLD_DEFAULT), which is also what happens at the first constructor call if nothing was installed.  0 on success,
 * -1 if a symbol is missing (RSGPU_Iterators_LastError names it). */
int RSGPU_Iterators_SetResultAPI(const RSGPU_ResultAPI *api, void *dl_handle);
const char *RSGPU_Iterators_LastError(void);

/* ---- constructors --------------------------------------------------------------------------------------------- */
/* One term of the query: its posting list on the device and what NewInvIndIterator_TermQuery receives besides the
 * index -- the RSQueryTerm (ownership passes to the iterator's Term record, as in the reference; may be NULL) and
 * the term node's weight. */
typedef struct RSGPU_TermArg {
  RSGPU_Postings *postings;
  RSQueryTerm *term;
  double weight;
} RSGPU_TermArg;

/* NewIntersectionIterator(its, num, max_slop, in_order, weight), iterators_ffi.h:309: documents in all `num` lists
 * (1..32), with max_slop >= 0 / in_order the proximity check of intersection.rs:205-215 on the device.  An empty
 * result gives an iterator that is at EOF from the first Read (the reference returns its Empty iterator).
 * The posting lists must outlive the iterator.  NULL on failure (RSGPU_Iterators_LastError / RSGPU_LastError). */
QueryIterator *RSGPU_NewIntersectionIterator(const RSGPU_TermArg *terms, size_t num, int32_t max_slop, bool in_order,
                                             double weight);
/* NewUnionIterator over term children (union_flat.rs, full mode): documents in any list; `current` holds the
 * children that matched the document, in the order of `terms`. */
QueryIterator *RSGPU_NewUnionIterator(const RSGPU_TermArg *terms, size_t num, double weight);
/* NewNotIterator (not.rs / not_optimized.rs): doc ids 1..max_doc_id the child does not hold -- with `universe`
 * (the existing-documents list) only those the universe holds; `current` is a Virtual result with
 * RS_FIELDMASK_ALL. */
QueryIterator *RSGPU_NewNotIterator(RSGPU_Postings *child, RSGPU_Postings *universe, t_docId max_doc_id, double weight);
/* Any hit list (RSGPU_Intersect / _IntersectEx / _Union / _EvalTree / _EvalTreeNodes) behind the vtable; `terms[i]` belongs to list i
 * of the call that built the hits.  With own_hits the iterator frees the hit list. */
QueryIterator *RSGPU_NewHitsIterator(RSGPU_Hits *hits, const RSGPU_TermArg *terms, size_t num, double weight,
                                     bool own_hits);
/* A two-level query tree behind one iterator (RSGPU_EvalTree, include/rsgpu_search.h): the root AND / OR over groups that
 * are a term, or the OR / AND of several terms -- e.g. the stemmer's (run|running|ran) (shoe|shoes).  `terms[i]` belongs
 * to q->lists[i] (q->lists itself is ignored: the posting lists are taken from `terms`).  `current` is the tree the
 * reference's nested iterators build: Intersection{Union{Term..}, Term, ..} with the groups in iteration order and, under
 * a union, only the children that matched the document. */
QueryIterator *RSGPU_NewTreeIterator(const RSGPU_TreeQuery *q, const RSGPU_TermArg *terms, double weight);
/* The same for a query tree of ANY depth (RSGPU_EvalTreeNodes): `nodes` in post-order, `terms[i]` belongs to list i. */
QueryIterator *RSGPU_NewTreeNodesIterator(const RSGPU_TreeNode *nodes, size_t n_nodes, const RSGPU_TermArg *terms, size_t num,
                                          double weight);
/* The hit list behind an iterator made here (score it in one batch with RSGPU_Hits_Score, re-rank with
 * RSGPU_Hits_KnnRerank, ...); NULL for foreign iterators. */
RSGPU_Hits *RSGPU_Iterator_Hits(QueryIterator *it);
/* Failure behaviour: a constructor returns NULL (RSGPU_Iterators_LastError says why; the hit list it made is released, and so
 * are the Term records -- with their terms -- if the module's allocator failed after they were handed over).  Read / SkipTo
 * return ITERATOR_TIMEOUT -- the one status that leaves an iterator where it was (iterator_api.h:100-102) -- when paging a
 * block of records from the device fails or the host runs out of memory; the same call may be repeated. */
/* Block size (hits) of the paging between device and host; default 65536.  Applies to iterators created later. */
void RSGPU_Iterators_SetBlock(size_t hits);

#ifdef __cplusplus
}
#endif
#endif /* RSGPU_RS_ITERATOR_H */
