/*
 * oracle/hybrid_harness.c -- runs the REFERENCE's own src/iterators/hybrid_reader.c (the HybridIterator: mode
 * selection, batches loop + merge-join, policy review, ad-hoc brute force, the K-bounded min-max heap) on top of a
 * VecSim library, without Redis.  TEST INFRASTRUCTURE ONLY, compiled only where /root/reference exists
 * (`make -C oracle ref` -> oracle/_ref/libref_hybrid_reader.so, which travels to the GPU box): it includes the
 * reference's headers and is linked with the reference's hybrid_reader.c and util/minmax_heap.c compiled IN PLACE.
 * The VecSim symbols stay undefined: the test loads redisearch_amd/lib/libVectorSimilarity.so RTLD_GLOBAL first, so
 * the reference's iterator drives the MI355X engine through the very seam RediSearch uses (SURVEY.md 8a rows
 * a1/a3/a4/a6/a8/a14, 8f-2).
 *
 * What this file supplies is what is Rust (or far away) in the real module:
 *   - the RSIndexResult constructors / accessors hybrid_reader.c calls (NewMetricResult, NewHybridResult,
 *     AggregateResult_AddChild, IndexResult_DeepCopy/Free/NumValue/SetNumValue, ...), over the reference's struct
 *     layout, an aggregate's records being a small heap array;
 *   - the child iterator: a sorted doc-id list behind the reference's QueryIterator vtable (Read / SkipTo / Rewind /
 *     NumEstimated), i.e. what NewSortedIdListIterator gives the pipeline;
 *   - no-op metrics / profiling hooks, VecSimType_sizeof, VecSimSearchMode_ToString;
 *   - xhr_run(): builds HybridIteratorParams as src/vector_index.c:262-290 does, creates the iterator with
 *     NewHybridVectorIterator, reads it to EOF, returns ids / distances / final search mode / batch statistics.
 */
#define REDISMODULE_MAIN /* defines the RedisModule_* API pointers here (all NULL: RS_IsMock, no timeouts) */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "hybrid_reader.h"
#include "VecSim/vec_sim.h"
#include "iterator_api.h"
#include "index_result_rs.h"
#include "types_ffi.h"
#include "metrics_ffi.h"
#include "iterators_ffi.h"
#include "search_ctx.h"
#include "spec.h"
#include "rmalloc.h"

#define XH_API __attribute__((visibility("default")))

/* ---- RSIndexResult plumbing ----------------------------------------------------------------------------------------- */
typedef struct {
  size_t len, cap;
  RSIndexResult **items;
} XRecs;

static XRecs *recs_of(const RSAggregateResult *a) { return (XRecs *)a->owned.records.ptr; }

static RSIndexResult *new_result(uint8_t tag) {
  RSIndexResult *r = calloc(1, sizeof *r);
  r->data.tag = tag;
  r->weight = 1.0;
  return r;
}
XH_API RSIndexResult *NewMetricResult(void) { return new_result(RSResultData_Metric); }
XH_API RSIndexResult *NewVirtualResult(double weight, t_fieldMask fieldMask) {
  RSIndexResult *r = new_result(RSResultData_Virtual);
  r->weight = weight;
  r->fieldMask = fieldMask;
  return r;
}
XH_API RSIndexResult *NewHybridResult(void) {
  RSIndexResult *r = new_result(RSResultData_HybridMetric);
  XRecs *x = calloc(1, sizeof *x);
  x->cap = 2;
  x->items = calloc(x->cap, sizeof *x->items);
  r->data.hybridmetric.owned.records.ptr = (void *)x;
  return r;
}
static int is_agg(const RSIndexResult *r) {
  return r->data.tag == RSResultData_Union || r->data.tag == RSResultData_Intersection ||
         r->data.tag == RSResultData_HybridMetric;
}
XH_API const RSAggregateResult *IndexResult_AggregateRef(const RSIndexResult *r) { return is_agg(r) ? &r->data.union_ : NULL; }
XH_API const RSAggregateResult *IndexResult_AggregateRefUnchecked(const RSIndexResult *r) { return &r->data.union_; }
XH_API RSAggregateResult *IndexResult_AggregateRefMutUnchecked(RSIndexResult *r) { return &r->data.union_; }
XH_API const RSIndexResult *AggregateResult_Get(const RSAggregateResult *a, size_t i) {
  XRecs *x = recs_of(a);
  return (x && i < x->len) ? x->items[i] : NULL;
}
XH_API const RSIndexResult *AggregateResult_GetUnchecked(const RSAggregateResult *a, size_t i) { return recs_of(a)->items[i]; }
XH_API RSIndexResult *AggregateResult_GetMutUnchecked(RSAggregateResult *a, size_t i) { return recs_of(a)->items[i]; }
XH_API void AggregateResult_AddChild(RSIndexResult *parent, RSIndexResult *child) {
  XRecs *x = recs_of(&parent->data.union_);
  if (x->len == x->cap) {
    x->cap *= 2;
    x->items = realloc(x->items, x->cap * sizeof *x->items);
  }
  x->items[x->len++] = child;
  parent->docId = child->docId; /* the aggregate takes its children's doc id, frequency and field mask */
  parent->freq += child->freq;
  parent->fieldMask |= child->fieldMask;
}
/* what an iterator implemented in C (redisearch_amd/csrc/query_iterators.c) builds its `current` with: aggregates that
 * BORROW their children (types_ffi.h:331,358), term records (:352), the offsets slice (:444, borrowed bytes) */
static RSIndexResult *new_borrowed_agg(uint8_t tag, size_t cap, double weight) {
  RSIndexResult *r = new_result(tag);
  XRecs *x = calloc(1, sizeof *x);
  x->cap = cap ? cap : 1;
  x->items = calloc(x->cap, sizeof *x->items);
  r->data.union_.owned.records.ptr = (void *)x;
  r->data.union_.tag = RSAggregateResult_Borrowed;
  r->weight = weight;
  return r;
}
XH_API RSIndexResult *NewIntersectResult(size_t cap, double weight) { return new_borrowed_agg(RSResultData_Intersection, cap, weight); }
XH_API RSIndexResult *NewUnionResult(size_t cap, double weight) { return new_borrowed_agg(RSResultData_Union, cap, weight); }
XH_API RSIndexResult *NewTokenRecord(struct RSQueryTerm *term, double weight) {
  RSIndexResult *r = new_result(RSResultData_Term);
  r->data.term.borrowed.term = term;
  r->weight = weight;
  return r;
}
XH_API void IndexResult_AggregateReset(RSIndexResult *r) {
  if (!is_agg(r)) return;
  XRecs *x = recs_of(&r->data.union_);
  if (r->data.union_.tag == RSAggregateResult_Owned)
    for (size_t i = 0; i < x->len; i++) IndexResult_Free(x->items[i]);
  x->len = 0;
}
XH_API void RSOffsetVector_SetData(RSOffsetSlice *offsets, const char *data, uint32_t len) {
  RSOffsetVector *v = (RSOffsetVector *)offsets;
  memcpy(&v->data, &data, sizeof data);
  v->len = len;
}
XH_API double IndexResult_NumValue(const RSIndexResult *r) { return r->data.numeric; }
XH_API void IndexResult_SetNumValue(RSIndexResult *r, double v) { r->data.numeric = v; }
XH_API void IndexResult_Free(RSIndexResult *r) {
  if (!r) return;
  if (is_agg(r)) {
    XRecs *x = recs_of(&r->data.union_);
    if (x) {
      if (r->data.union_.tag == RSAggregateResult_Owned)
        for (size_t i = 0; i < x->len; i++) IndexResult_Free(x->items[i]);
      free(x->items);
      free(x);
    }
  }
  free(r);
}
XH_API RSIndexResult *IndexResult_DeepCopy(const RSIndexResult *src) {
  RSIndexResult *r = malloc(sizeof *r);
  memcpy(r, src, sizeof *r);
  r->metrics.ptr = NULL;
  if (is_agg(src)) {
    const XRecs *sx = recs_of(&src->data.union_);
    XRecs *x = calloc(1, sizeof *x);
    x->cap = sx && sx->len ? sx->len : 2;
    x->items = calloc(x->cap, sizeof *x->items);
    if (sx)
      for (size_t i = 0; i < sx->len; i++) x->items[x->len++] = IndexResult_DeepCopy(sx->items[i]);
    r->data.union_.owned.records.ptr = (void *)x;
    r->data.union_.tag = RSAggregateResult_Owned;
  }
  return r;
}

/* yieldable metrics, profiling, TTL: not on this path */
XH_API void ResultMetrics_Add(RSIndexResult *r, const RLookupKey *key, double value) { (void)r, (void)key, (void)value; }
XH_API void ResultMetrics_Reset(RSIndexResult *r) { (void)r; }
XH_API void MetricsVec_UpdateValue(MetricsVec *m, const RLookupKey *key, double value) { (void)m, (void)key, (void)value; }
XH_API void RSYieldableMetric_Concat(MetricsVec *dst, MetricsVec *src) { (void)dst, (void)src; }
XH_API bool IsWildcardIterator(const QueryIterator *it) { return it->type == WILDCARD_ITERATOR; }
XH_API QueryIterator *IntoProfiled(QueryIterator *it) { return it; }
XH_API void Hybrid_PrintProfile(const QueryIterator *it, RsMapBuilder *map, RsProfilePrintCtx *ctx) { (void)it, (void)map, (void)ctx; }
XH_API bool TimeToLiveTable_FieldSatisfiesPredicate(const struct TimeToLiveTable *table, t_docId doc_id, uint16_t field_index,
                                                    enum FieldExpirationPredicate predicate,
                                                    const t_expirationTimePoint *expiration_point) {
  (void)table, (void)doc_id, (void)field_index, (void)predicate, (void)expiration_point;
  return true;
}

/* reference src/vector_index.c (VecSimType_sizeof, VecSimSearchMode_ToString): restated, that file needs the module */
XH_API size_t VecSimType_sizeof(VecSimType type) {
  switch (type) {
    case VecSimType_FLOAT64: case VecSimType_INT64: return 8;
    case VecSimType_FLOAT32: case VecSimType_INT32: return 4;
    case VecSimType_BFLOAT16: case VecSimType_FLOAT16: return 2;
    default: return 1;
  }
}
XH_API const char *VecSimSearchMode_ToString(VecSearchMode m) {
  static const char *names[] = {"EMPTY_MODE", "STANDARD_KNN", "HYBRID_ADHOC_BF", "HYBRID_BATCHES",
                                "HYBRID_BATCHES_TO_ADHOC_BF", "RANGE_QUERY"};
  return (unsigned)m < 6 ? names[m] : "?";
}

/* ---- the child: a sorted doc-id list behind the QueryIterator vtable --------------------------------------------------- */
typedef struct {
  QueryIterator base;
  t_docId *ids;
  size_t n, pos, estimate;
  RSIndexResult *res;
  size_t reads, skips, rewinds;
} IdList;

static IteratorStatus il_yield(IdList *it) {
  if (it->pos >= it->n) {
    it->base.atEOF = true;
    it->base.current = NULL;
    return ITERATOR_EOF;
  }
  it->base.lastDocId = it->res->docId = it->ids[it->pos++];
  it->base.current = it->res;
  return ITERATOR_OK;
}
static IteratorStatus il_read(QueryIterator *b) {
  IdList *it = (IdList *)b;
  it->reads++;
  return il_yield(it);
}
static IteratorStatus il_skipto(QueryIterator *b, t_docId id) {
  IdList *it = (IdList *)b;
  it->skips++;
  size_t lo = it->pos, hi = it->n;
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    if (it->ids[mid] < id) lo = mid + 1;
    else hi = mid;
  }
  it->pos = lo;
  IteratorStatus s = il_yield(it);
  return (s == ITERATOR_OK && b->lastDocId != id) ? ITERATOR_NOTFOUND : s;
}
static void il_rewind(QueryIterator *b) {
  IdList *it = (IdList *)b;
  it->rewinds++;
  it->pos = 0;
  b->lastDocId = 0;
  b->atEOF = false;
  b->current = NULL;
}
static size_t il_estimated(const QueryIterator *b) { return ((const IdList *)b)->estimate; }
static void il_free(QueryIterator *b) {
  IdList *it = (IdList *)b;
  free(it->ids);
  IndexResult_Free(it->res);
  free(it);
}
static QueryIterator *new_id_list(const uint64_t *ids, size_t n, size_t estimate) {
  IdList *it = calloc(1, sizeof *it);
  it->ids = malloc((n ? n : 1) * sizeof *it->ids);
  for (size_t i = 0; i < n; i++) it->ids[i] = ids[i];
  it->n = n;
  it->estimate = estimate ? estimate : n;
  it->res = NewVirtualResult(1.0, RS_FIELDMASK_ALL);
  it->res->freq = 1;
  it->base.type = ID_LIST_SORTED_ITERATOR;
  it->base.NumEstimated = il_estimated;
  it->base.Read = il_read;
  it->base.SkipTo = il_skipto;
  it->base.Rewind = il_rewind;
  it->base.Free = il_free;
  it->base.Revalidate = Default_Revalidate;
  return &it->base;
}

/* ---- the driver -------------------------------------------------------------------------------------------------------- */
typedef struct {
  int search_mode_in;     /* 0 auto, VECSIM_HYBRID_ADHOC_BF / VECSIM_HYBRID_BATCHES to force (HYBRID_POLICY) */
  size_t batch_size;      /* BATCH_SIZE, 0 = the iterator's own sizing */
  int can_trim;           /* canTrimDeepResults: Metric results instead of HybridMetric(vector, child) */
  int read_twice;         /* Rewind after EOF and read again (rewind KATs) */
  size_t child_estimate;  /* NumEstimated of the child (an upper bound the query planner hands over); 0 = its length */
  /* out */
  int search_mode_out;
  size_t num_iterations, max_batch_size, child_reads, child_skips, child_rewinds;
  int timed_out;
  int second_pass_identical;
} XhrOpts;

static long read_all(QueryIterator *it, uint64_t *ids_out, double *scores_out, size_t cap, int *timed_out) {
  long n = 0;
  IteratorStatus s;
  while ((s = it->Read(it)) == ITERATOR_OK) {
    const RSIndexResult *r = it->current;
    double d = r->data.tag == RSResultData_Metric
                   ? IndexResult_NumValue(r)
                   : IndexResult_NumValue(AggregateResult_GetUnchecked(IndexResult_AggregateRefUnchecked(r), 0));
    if ((size_t)n < cap) {
      ids_out[n] = r->docId;
      scores_out[n] = d;
    }
    n++;
  }
  if (s == ITERATOR_TIMEOUT) *timed_out = 1;
  return n;
}

/* child_ids == NULL: pure KNN (no child).  Results come in the order the iterator yields them: ascending distance
 * for STANDARD_KNN, heap pop-min order (ascending distance, ties: see cmpVecSimResByScore) for the hybrid modes. */
static long run_with(VecSimIndex *index, int vtype, int metric, size_t dim, const void *query_blob, size_t k,
                     QueryIterator *child_it, IdList *child, XhrOpts *o, uint64_t *ids_out, double *scores_out, size_t cap) {
  static IndexSpec spec;      /* zeroed: no diskSpec, no TTL table */
  static RedisSearchCtx sctx;
  memset(&spec, 0, sizeof spec);
  memset(&sctx, 0, sizeof sctx);
  sctx.spec = &spec;
  FieldFilterContext filter;
  memset(&filter, 0, sizeof filter);
  filter.field.index_tag = FieldMaskOrIndex_Index;
  filter.field.index = RS_INVALID_FIELD_INDEX;
  HybridIteratorParams hp;
  memset(&hp, 0, sizeof hp);
  hp.sctx = &sctx;
  hp.index = index;
  hp.dim = dim;
  hp.elementType = (VecSimType)vtype;
  hp.spaceMetric = (VecSimMetric)metric;
  hp.query.vector = (void *)query_blob;
  hp.query.vecLen = dim * VecSimType_sizeof((VecSimType)vtype);
  hp.query.k = k;
  hp.query.order = BY_SCORE;
  hp.qParams.searchMode = (VecSearchMode)o->search_mode_in;
  hp.qParams.batchSize = o->batch_size;
  hp.canTrimDeepResults = o->can_trim != 0;
  hp.childIt = child_it;
  hp.filterCtx = &filter;
  hp.timeout.tv_sec = (time_t)1 << 40;
  QueryError status;
  memset(&status, 0, sizeof status);
  QueryIterator *it = NewHybridVectorIterator(hp, &status);
  if (!it) return -1;
  if (it == hp.childIt) { /* reduced away (empty child) */
    it->Free(it);
    return 0;
  }
  long n = read_all(it, ids_out, scores_out, cap, &o->timed_out);
  const HybridIterator *hi = (const HybridIterator *)it;
  o->search_mode_out = (int)hi->searchMode;
  o->num_iterations = hi->numIterations;
  o->max_batch_size = hi->maxBatchSize;
  if (o->read_twice) {
    uint64_t *ids2 = malloc((cap ? cap : 1) * sizeof *ids2);
    double *sc2 = malloc((cap ? cap : 1) * sizeof *sc2);
    it->Rewind(it);
    int to = 0;
    long n2 = read_all(it, ids2, sc2, cap, &to);
    size_t m = (size_t)(n < (long)cap ? n : (long)cap);
    o->second_pass_identical = n2 == n && !memcmp(ids2, ids_out, m * sizeof *ids2) && !memcmp(sc2, scores_out, m * sizeof *sc2);
    free(ids2);
    free(sc2);
  }
  if (child) {
    o->child_reads = child->reads;
    o->child_skips = child->skips;
    o->child_rewinds = child->rewinds;
  }
  it->Free(it);
  return n;
}

XH_API long xhr_run(VecSimIndex *index, int vtype, int metric, size_t dim, const void *query_blob, size_t k,
                    const uint64_t *child_ids, size_t n_child, XhrOpts *o, uint64_t *ids_out, double *scores_out,
                    size_t cap) {
  QueryIterator *c = child_ids ? new_id_list(child_ids, n_child, o->child_estimate) : NULL;
  return run_with(index, vtype, metric, dim, query_blob, k, c, (IdList *)c, o, ids_out, scores_out, cap);
}
/* The same with a child iterator made elsewhere -- any implementation of the reference's vtable, e.g. the product's
 * GPU-backed intersection (include/rs_iterator.h).  The HybridIterator owns and frees it (child->Free). */
XH_API long xhr_run_child(VecSimIndex *index, int vtype, int metric, size_t dim, const void *query_blob, size_t k,
                          QueryIterator *child, XhrOpts *o, uint64_t *ids_out, double *scores_out, size_t cap) {
  return run_with(index, vtype, metric, dim, query_blob, k, child, NULL, o, ids_out, scores_out, cap);
}
