/*
 * oracle/ext_harness.c -- a stand-in for the RediSearch MODULE side of the extension API, so that scorer extensions
 * can be loaded and driven without Redis (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).
 *
 * It plays three parts of the module:
 *   1. the extension registry and loader -- what src/extension.c:67-145 does: RegisterScoringFunction refuses a
 *      NULL function and a duplicate alias (REDISEARCH_ERR), keeps its own copy of the alias; an extension is
 *      dlopen()ed RTLD_LOCAL and its init symbol called with an RSExtensionCtx.  The same loader takes
 *        - redisearch_amd/lib/librsgpu_scorers.so, init symbol "RS_ExtensionInit"  (the product under test), and
 *        - oracle/_ref/libref_default_ext.so, init symbol "DefaultExtensionInit"  (the REFERENCE's own
 *          src/ext/default.c + src/index_result/index_result.c compiled in place by `make -C oracle ref`);
 *   2. the result-tree accessors that are Rust in the real module (src/redisearch_rs/headers/types_ffi.h,
 *      query_term_ffi.h), here over plain C structs: this library is loaded RTLD_GLOBAL, so both extensions bind
 *      their undefined IndexResult_* / AggregateResult_* / QueryTerm_* symbols to these;
 *   3. the caller rpscoreNext (src/result_processor.c:570-603): builds ScoringFunctionArgs (+ an RSScoreExplain root
 *      when asked), calls the scorer, reads args.scrExp back, renders the explanation tree as indented text.
 *
 * Result trees are laid out with include/rs_extension.h's structs, whose layout equals the reference's
 * (tests/test_scorer_plugin.py::test_layout_matches_reference_headers), so the reference's compiled code reads
 * them directly (r->freq, r->weight, r->data.tag).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rs_extension.h"
#include "rs_iterator.h" /* QueryIterator: the iterator drivers at the end of this file */

#define XH_API __attribute__((visibility("default")))

/* ---- what a "query term" and an aggregate's record vector are in this harness ---------------------------------- */
struct RSQueryTerm {
  double idf, bm25_idf;
  char *str;
  size_t len;
};

typedef struct {
  size_t len, cap;
  RSIndexResult *items[];
} XVec;
enum { XAGG_BORROWED = 0, XAGG_OWNED = 1 }; /* RSAggregateResult_Borrowed / _Owned, index_result_rs.h:486-487 */

static char g_err[512];
XH_API const char *xh_last_error(void) { return g_err; }

/* ---- accessors (module side) -------------------------------------------------------------------------------------- */
XH_API const RSAggregateResult *IndexResult_AggregateRefUnchecked(const RSIndexResult *r) { return &r->data.agg; }
XH_API const RSAggregateResult *IndexResult_AggregateRef(const RSIndexResult *r) {
  return (r->data.tag & (RSResultData_Union | RSResultData_Intersection | RSResultData_HybridMetric)) ? &r->data.agg : NULL;
}
XH_API size_t AggregateResult_NumChildren(const RSAggregateResult *a) { return ((const XVec *)a->records)->len; }
XH_API AggregateRecordsSlice AggregateResult_GetRecordsSlice(const RSAggregateResult *a) {
  const XVec *v = (const XVec *)a->records;
  AggregateRecordsSlice s = {(const RSIndexResult *const *)v->items, v->len};
  return s;
}
XH_API const RSIndexResult *AggregateResult_Get(const RSAggregateResult *a, size_t i) {
  const XVec *v = (const XVec *)a->records;
  return i < v->len ? v->items[i] : NULL;
}
XH_API const RSIndexResult *AggregateResult_GetUnchecked(const RSAggregateResult *a, size_t i) {
  return ((const XVec *)a->records)->items[i];
}
XH_API uint8_t AggregateResult_KindMask(const RSAggregateResult *a) { return a->kind_mask; }
XH_API RSQueryTerm *IndexResult_QueryTermRef(const RSIndexResult *r) {
  return r->data.tag == RSResultData_Term ? (RSQueryTerm *)r->data.term.term : NULL;
}
XH_API double QueryTerm_GetIDF(const RSQueryTerm *t) { return t->idf; }
XH_API double QueryTerm_GetBM25_IDF(const RSQueryTerm *t) { return t->bm25_idf; }
XH_API const char *QueryTerm_GetStrAndLen(const RSQueryTerm *t, size_t *n) {
  if (n) *n = t->len;
  return t->str;
}

/* offsets: in this harness a term's "encoded offsets" are a plain uint32 array (positions ascending);
 * data = the array, len = number of positions.  An aggregate iterates the ascending merge of its terms' positions. */
typedef struct {
  const uint8_t *data;
  uint32_t len;
} XOffsetSlice;
XH_API const XOffsetSlice *IndexResult_TermOffsetsRef(const RSIndexResult *r) {
  return r->data.tag == RSResultData_Term ? (const XOffsetSlice *)&r->data.term.offsets : NULL;
}
XH_API uint32_t RSOffsetVector_Len(const XOffsetSlice *s) { return s ? s->len : 0; }

typedef struct RSOffsetIterator { /* reference src/redisearch.h:195-200 */
  void *ctx;
  uint32_t (*Next)(void *ctx, RSQueryTerm **term);
  void (*Rewind)(void *ctx);
  void (*Free)(void *ctx);
} RSOffsetIterator;

typedef struct {
  uint32_t *pos;
  size_t n, i;
} XOffIt;
static uint32_t xoff_next(void *c, RSQueryTerm **t) {
  XOffIt *it = (XOffIt *)c;
  if (t) *t = NULL;
  return it->i < it->n ? it->pos[it->i++] : UINT32_MAX; /* RS_OFFSETVECTOR_EOF */
}
static void xoff_rewind(void *c) { ((XOffIt *)c)->i = 0; }
static void xoff_free(void *c) {
  free(((XOffIt *)c)->pos);
  free(c);
}
static size_t count_positions(const RSIndexResult *r) {
  if (r->data.tag == RSResultData_Term) return r->data.term.offsets.len;
  if (!IndexResult_AggregateRef(r)) return 0;
  size_t n = 0;
  const XVec *v = (const XVec *)r->data.agg.records;
  for (size_t i = 0; i < v->len; i++) n += count_positions(v->items[i]);
  return n;
}
static size_t gather_positions(const RSIndexResult *r, uint32_t *out) {
  if (r->data.tag == RSResultData_Term) {
    memcpy(out, r->data.term.offsets.data, (size_t)r->data.term.offsets.len * 4);
    return r->data.term.offsets.len;
  }
  if (!IndexResult_AggregateRef(r)) return 0;
  size_t n = 0;
  const XVec *v = (const XVec *)r->data.agg.records;
  for (size_t i = 0; i < v->len; i++) n += gather_positions(v->items[i], out + n);
  return n;
}
static int cmp_u32(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : x > y;
}
XH_API RSOffsetIterator RSIndexResult_IterateOffsets(const RSIndexResult *r) {
  XOffIt *it = (XOffIt *)calloc(1, sizeof *it);
  it->n = count_positions(r);
  it->pos = (uint32_t *)malloc((it->n ? it->n : 1) * 4);
  gather_positions(r, it->pos);
  qsort(it->pos, it->n, 4, cmp_u32);
  RSOffsetIterator o = {it, xoff_next, xoff_rewind, xoff_free};
  return o;
}

/* the module's explain() (src/score_explain.c:54-63), for the reference's EXPLAIN macro */
XH_API void explain(RSScoreExplain *e, char *fmt, ...) {
  char *old = e->str;
  va_list ap;
  va_start(ap, fmt);
  if (vasprintf(&e->str, fmt, ap) < 0) e->str = NULL;
  va_end(ap);
  free(old);
}

/* ---- tree construction ----------------------------------------------------------------------------------------------- */
XH_API RSIndexResult *xh_term(double weight, uint32_t freq, int has_term, double idf, double bm25_idf, const char *str,
                              const uint32_t *offsets, size_t n_offsets) {
  RSIndexResult *r = (RSIndexResult *)calloc(1, sizeof *r);
  r->data.tag = RSResultData_Term;
  r->weight = weight;
  r->freq = freq;
  if (has_term) {
    RSQueryTerm *t = (RSQueryTerm *)calloc(1, sizeof *t);
    t->idf = idf;
    t->bm25_idf = bm25_idf;
    t->str = strdup(str ? str : "");
    t->len = strlen(t->str);
    r->data.term.term = t;
  }
  if (n_offsets) {
    uint32_t *p = (uint32_t *)malloc(n_offsets * 4);
    memcpy(p, offsets, n_offsets * 4);
    r->data.term.offsets.data = (const uint8_t *)p;
    r->data.term.offsets.len = (uint32_t)n_offsets;
  }
  return r;
}

XH_API RSIndexResult *xh_leaf(int tag, double weight, uint32_t freq, double num) {
  RSIndexResult *r = (RSIndexResult *)calloc(1, sizeof *r);
  r->data.tag = (uint8_t)tag;
  r->weight = weight;
  r->freq = freq;
  if (tag == RSResultData_Numeric || tag == RSResultData_Metric) r->data.num = num;
  return r;
}

/* Takes ownership of the children; freq and kind mask accumulate as AggregateResult_AddChild does. */
XH_API RSIndexResult *xh_agg(int tag, double weight, RSIndexResult **kids, size_t n) {
  RSIndexResult *r = (RSIndexResult *)calloc(1, sizeof *r);
  XVec *v = (XVec *)calloc(1, sizeof *v + (n ? n : 1) * sizeof(RSIndexResult *));
  v->len = n;
  v->cap = n ? n : 1;
  r->data.agg.tag = XAGG_OWNED;
  r->data.tag = (uint8_t)tag;
  r->weight = weight;
  for (size_t i = 0; i < n; i++) {
    v->items[i] = kids[i];
    r->freq += kids[i]->freq;
    r->data.agg.kind_mask |= kids[i]->data.tag;
  }
  r->data.agg.records = v;
  return r;
}

XH_API void xh_free(RSIndexResult *r) {
  if (!r) return;
  if (r->data.tag == RSResultData_Term) {
    RSQueryTerm *t = (RSQueryTerm *)r->data.term.term;
    if (t) {
      free(t->str);
      free(t);
    }
    free((void *)r->data.term.offsets.data);
  } else if (IndexResult_AggregateRef(r)) {
    XVec *v = (XVec *)r->data.agg.records;
    if (r->data.agg.tag == XAGG_OWNED)
      for (size_t i = 0; i < v->len; i++) xh_free(v->items[i]);
    free(v);
  }
  free(r);
}

/* ---- the module's RSIndexResult constructors (src/redisearch_rs/headers/types_ffi.h:89,233,263,331,352,358,364,444):
 * what an iterator implemented in C builds its `current` with.  Aggregates made here BORROW their children
 * (RSIndexResult::build_intersect / build_union), exactly like the reference's. ------------------------------------------ */
/* fault injection for the iterator library's error paths: the (n+1)-th constructor call from now on returns NULL once
 * (the module's allocator never does -- RedisModule_Alloc aborts -- but a library must not crash or leak if it did) */
static long g_fail_after = -1;
XH_API void xh_fail_constructor_after(long n) { g_fail_after = n; }
static int failing(void) { return g_fail_after >= 0 && g_fail_after-- == 0; }

static RSIndexResult *new_agg(int tag, size_t cap, double weight) {
  if (failing()) return NULL;
  RSIndexResult *r = (RSIndexResult *)calloc(1, sizeof *r);
  XVec *v = (XVec *)calloc(1, sizeof *v + (cap ? cap : 1) * sizeof(RSIndexResult *));
  v->cap = cap ? cap : 1;
  r->data.tag = (uint8_t)tag;
  r->data.agg.tag = XAGG_BORROWED;
  r->data.agg.records = v;
  r->weight = weight;
  return r;
}
XH_API RSIndexResult *NewIntersectResult(size_t cap, double weight) { return new_agg(RSResultData_Intersection, cap, weight); }
XH_API RSIndexResult *NewUnionResult(size_t cap, double weight) { return new_agg(RSResultData_Union, cap, weight); }
XH_API RSIndexResult *NewVirtualResult(double weight, t_fieldMask field_mask) {
  if (failing()) return NULL;
  RSIndexResult *r = (RSIndexResult *)calloc(1, sizeof *r); /* RawIndexResultBuilder::virt: doc 0, freq 0 */
  r->data.tag = RSResultData_Virtual;
  r->weight = weight;
  r->fieldMask = field_mask;
  return r;
}
/* types_ffi/src/lib.rs:105-122: a term record with frequency 0, field mask 0, the term owned by the record */
XH_API RSIndexResult *NewTokenRecord(RSQueryTerm *term, double weight) {
  if (failing()) { /* the term was handed over: it goes with the failed record */
    if (term) free(term->str), free(term);
    return NULL;
  }
  RSIndexResult *r = (RSIndexResult *)calloc(1, sizeof *r);
  r->data.tag = RSResultData_Term;
  r->data.term.term = term;
  r->weight = weight;
  return r;
}
/* RSIndexResult::push_borrowed (index_result/src/core/mod.rs:985-1010): the parent takes the child's doc id, adds its
 * frequency, ORs its field mask; an owned parent takes the child itself */
XH_API void AggregateResult_AddChild(RSIndexResult *parent, RSIndexResult *child) {
  if (!IndexResult_AggregateRef(parent)) return;
  XVec *v = (XVec *)parent->data.agg.records;
  if (v->len == v->cap) {
    v->cap *= 2;
    v = (XVec *)realloc(v, sizeof *v + v->cap * sizeof(RSIndexResult *));
    parent->data.agg.records = v;
  }
  v->items[v->len++] = child;
  parent->docId = child->docId;
  parent->freq += child->freq;
  parent->fieldMask |= child->fieldMask;
  parent->data.agg.kind_mask |= child->data.tag;
}
XH_API void IndexResult_AggregateReset(RSIndexResult *r) { /* types_ffi.h:233: children vector and kind mask */
  if (!IndexResult_AggregateRef(r)) return;
  XVec *v = (XVec *)r->data.agg.records;
  if (r->data.agg.tag == XAGG_OWNED)
    for (size_t i = 0; i < v->len; i++) xh_free(v->items[i]);
  v->len = 0;
  r->data.agg.kind_mask = 0;
}
XH_API void IndexResult_Free(RSIndexResult *r) { xh_free(r); }
/* RSOffsetVector_SetData(offsets, data, len), types_ffi.h:444: `data` is the ENCODED offsets blob of a posting record
 * (varint deltas, reference src/redisearch_rs/varint/src/lib.rs; src/varint.h ReadVarint).  In this harness a term's
 * offsets are the decoded positions (the header of this file), so the blob is decoded here -- which is what the
 * module's offset iterator does lazily. */
XH_API void RSOffsetVector_SetData(void *offsets, const char *data, uint32_t len) {
  XOffsetSlice *s = (XOffsetSlice *)offsets;
  free((void *)s->data);
  s->data = NULL;
  s->len = 0;
  if (!data || !len) return;
  uint32_t *pos = (uint32_t *)malloc((size_t)len * 4); /* at most one position per byte */
  uint32_t n = 0, at = 0, last = 0;
  while (at < len) {
    unsigned char c = (unsigned char)data[at++];
    uint32_t val = c & 127;
    while ((c >> 7) && at < len) {
      ++val;
      c = (unsigned char)data[at++];
      val = (val << 7) | (c & 127);
    }
    last += val;
    pos[n++] = last;
  }
  s->data = (const uint8_t *)pos;
  s->len = n;
}
/* a query term for NewTokenRecord (NewQueryTerm + the idf pair Term::new computes, term.rs:91) */
XH_API RSQueryTerm *xh_new_term(double idf, double bm25_idf, const char *str) {
  RSQueryTerm *t = (RSQueryTerm *)calloc(1, sizeof *t);
  t->idf = idf;
  t->bm25_idf = bm25_idf;
  t->str = strdup(str ? str : "");
  t->len = strlen(t->str);
  return t;
}

/* ---- registry + loader (src/extension.c) ----------------------------------------------------------------------------- */
typedef struct {
  char *alias;
  RSScoringFunction fn;
  RSFreeFunction ff;
  void *privdata;
} XScorer;
static XScorer g_scorers[128];
static size_t g_nscorers, g_nexpanders;
static void *g_handles[16];
static size_t g_nhandles;
static int (*g_ref_slop)(const RSIndexResult *);

static XScorer *find_scorer(const char *alias) {
  for (size_t i = 0; i < g_nscorers; i++)
    if (!strcmp(g_scorers[i].alias, alias)) return &g_scorers[i]; /* case sensitive, like the TrieMap */
  return NULL;
}
static int reg_scorer(const char *alias, RSScoringFunction fn, RSFreeFunction ff, void *privdata) {
  if (!fn || find_scorer(alias) || g_nscorers == 128) return REDISEARCH_ERR;
  XScorer s = {strdup(alias), fn, ff, privdata};
  g_scorers[g_nscorers++] = s;
  return REDISEARCH_OK;
}
static int reg_expander(const char *alias, RSQueryTokenExpander exp, RSFreeFunction ff, void *privdata) {
  (void)alias, (void)ff, (void)privdata;
  if (!exp) return REDISEARCH_ERR;
  g_nexpanders++;
  return REDISEARCH_OK;
}

XH_API void xh_reset(void) {
  for (size_t i = 0; i < g_nscorers; i++) free(g_scorers[i].alias);
  g_nscorers = g_nexpanders = 0;
}
XH_API size_t xh_num_scorers(void) { return g_nscorers; }
XH_API size_t xh_num_expanders(void) { return g_nexpanders; }
XH_API const char *xh_alias(size_t i) { return i < g_nscorers ? g_scorers[i].alias : NULL; }

/* Extension_LoadDynamic: returns the init function's result (REDISEARCH_OK/ERR), -1 when dlopen fails, -2 when the
 * init symbol is missing.  now != 0 binds every symbol at load (RTLD_NOW, what the module does); the reference's
 * default.c also carries its query EXPANDERS, whose dependencies (stemmer, tokenizer, synonym map) are not part of
 * this path, so that one library is bound lazily. */
XH_API int xh_load(const char *path, const char *init_symbol, int now) {
  void *h = dlopen(path, (now ? RTLD_NOW : RTLD_LAZY) | RTLD_LOCAL);
  if (!h) {
    snprintf(g_err, sizeof g_err, "%s", dlerror());
    return -1;
  }
  RSExtensionInitFunc init = (RSExtensionInitFunc)dlsym(h, init_symbol);
  if (!init) {
    snprintf(g_err, sizeof g_err, "no %s in %s", init_symbol, path);
    dlclose(h);
    return -2;
  }
  if (g_nhandles < 16) g_handles[g_nhandles++] = h;
  void *slop = dlsym(h, "IndexResult_MinOffsetDelta");
  if (slop) g_ref_slop = (int (*)(const RSIndexResult *))slop;
  RSExtensionCtx ctx = {reg_scorer, reg_expander};
  return init(&ctx);
}

XH_API int xh_has_ref_slop(void) { return g_ref_slop != NULL; }
XH_API int xh_ref_slop(const RSIndexResult *r) { return g_ref_slop ? g_ref_slop(r) : -1; }

/* ---- the caller ------------------------------------------------------------------------------------------------------- */
static int g_fixed_slop;
static int fixed_slop(const RSIndexResult *r) {
  (void)r;
  return g_fixed_slop;
}

static void render(const RSScoreExplain *e, int depth, char **out, size_t *cap) {
  int n = snprintf(*out, *cap, "%*s%s\n", depth * 2, "", e->str ? e->str : "(null)");
  if (n > 0) {
    size_t adv = (size_t)n < *cap ? (size_t)n : (*cap ? *cap - 1 : 0);
    *out += adv;
    *cap -= adv;
  }
  for (int i = 0; i < e->numChildren; i++) render(&e->children[i], depth + 1, out, cap);
}
static void destroy(RSScoreExplain *e) { /* SEDestroy's recursion, src/score_explain.c:35-42 */
  for (int i = 0; i < e->numChildren; i++) destroy(&e->children[i]);
  free(e->children);
  free(e->str);
}

typedef struct {
  float doc_score;
  uint32_t max_term_freq, doc_len;
  const void *payload;
  size_t payload_len;
  size_t num_docs;
  double avg_doc_len;
  uint64_t tanh_factor;
  const void *qdata;
  size_t qdatalen;
  double min_score;
  int slop; /* > 0: GetSlop returns this; <= 0: the loaded reference's IndexResult_MinOffsetDelta */
} XScoreArgs;

/* Returns the score; NaN with xh_last_error() set when the alias is unknown or no slop source exists.  With
 * explain_out != NULL an RSScoreExplain root is handed to the scorer and rendered (two spaces per depth). */
XH_API double xh_score(const char *alias, const RSIndexResult *res, const XScoreArgs *a, char *explain_out, size_t cap) {
  const XScorer *s = find_scorer(alias);
  g_err[0] = 0;
  if (!s) {
    snprintf(g_err, sizeof g_err, "no scorer %s", alias);
    return 0.0 / 0.0;
  }
  if (a->slop <= 0 && !g_ref_slop) {
    snprintf(g_err, sizeof g_err, "no slop source");
    return 0.0 / 0.0;
  }
  RSPayload pl = {(char *)a->payload, a->payload_len};
  RSDocumentMetadata dmd;
  memset(&dmd, 0, sizeof dmd);
  dmd.score = a->doc_score;
  dmd.maxTermFreq = a->max_term_freq;
  dmd.docLen = a->doc_len;
  if (a->payload) {
    dmd.flags = RS_DOCUMENT_HAS_PAYLOAD;
    dmd.payload = &pl;
  }
  ScoringFunctionArgs args;
  memset(&args, 0, sizeof args);
  args.extdata = s->privdata;
  args.qdata = a->qdata;
  args.qdatalen = a->qdatalen;
  args.indexStats.numDocs = a->num_docs;
  args.indexStats.avgDocLen = a->avg_doc_len;
  args.tanhFactor = a->tanh_factor;
  g_fixed_slop = a->slop;
  args.GetSlop = a->slop > 0 ? fixed_slop : g_ref_slop;
  if (explain_out) args.scrExp = calloc(1, sizeof(RSScoreExplain));
  const double v = s->fn(&args, res, &dmd, a->min_score);
  if (explain_out) {
    RSScoreExplain *root = (RSScoreExplain *)args.scrExp;
    if (cap) explain_out[0] = 0;
    render(root, 0, &explain_out, &cap);
    destroy(root);
    free(root);
  }
  return v;
}

/* ---- iterator drivers: what the query pipeline does with a QueryIterator (src/result_processor.c rpQueryItNext reads,
 * composite iterators skip) -- here over any iterator with the reference's vtable, i.e. the product's
 * librsgpu_iterators.so --------------------------------------------------------------------------------------------- */
static uint64_t pos_hash(const RSIndexResult *t) { /* order-sensitive digest of a term record's positions */
  uint64_t h = 1469598103934665603ull;
  const uint32_t *p = (const uint32_t *)t->data.term.offsets.data;
  for (uint32_t i = 0; i < t->data.term.offsets.len; i++) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}
/* Read to EOF.  Per hit: doc id, the aggregate's frequency and field mask, its number of children; per hit and child
 * slot c < max_children (planes of `cap`): the child's doc id == the hit's (1/0), frequency, field mask (lo/hi),
 * number of positions and their digest.  Returns the number of hits (may exceed cap: the rest is counted only),
 * -1 on a status other than OK / EOF. */
XH_API long xh_iter_drain(QueryIterator *it, size_t cap, size_t max_children, uint64_t *ids, uint32_t *freq,
                          uint64_t *mask_lo, uint64_t *mask_hi, uint32_t *n_children, uint32_t *c_freq, uint64_t *c_mask_lo,
                          uint64_t *c_mask_hi, uint32_t *c_npos, uint64_t *c_hash, uint8_t *c_same_doc) {
  size_t n = 0;
  for (;;) {
    const IteratorStatus st = it->Read(it);
    if (st == ITERATOR_EOF) break;
    if (st != ITERATOR_OK || !it->current || it->current->docId != it->lastDocId) return -1;
    const RSIndexResult *r = it->current;
    if (n < cap) {
      ids[n] = r->docId;
      freq[n] = r->freq;
      mask_lo[n] = (uint64_t)r->fieldMask;
      mask_hi[n] = (uint64_t)(r->fieldMask >> 64);
      size_t k = 0;
      if (IndexResult_AggregateRef(r)) {
        const XVec *v = (const XVec *)r->data.agg.records;
        k = v->len;
        for (size_t c = 0; c < k && c < max_children; c++) {
          const RSIndexResult *t = v->items[c];
          c_freq[c * cap + n] = t->freq;
          c_mask_lo[c * cap + n] = (uint64_t)t->fieldMask;
          c_mask_hi[c * cap + n] = (uint64_t)(t->fieldMask >> 64);
          c_npos[c * cap + n] = t->data.tag == RSResultData_Term ? t->data.term.offsets.len : 0;
          c_hash[c * cap + n] = t->data.tag == RSResultData_Term ? pos_hash(t) : 0;
          c_same_doc[c * cap + n] = t->docId == r->docId;
        }
      }
      n_children[n] = (uint32_t)k;
    }
    n++;
  }
  if (!it->atEOF || it->current) return -1; /* iterator_api.h:96-99 */
  return (long)n;
}
/* Read to EOF the way a lean consumer does (rpscoreNext with a trivial scorer): per hit, touch the aggregate and every
 * child record once.  Returns the number of hits, -1 on a protocol violation; *checksum folds doc ids and frequencies. */
XH_API long xh_iter_drain_lean(QueryIterator *it, uint64_t *checksum) {
  size_t n = 0;
  uint64_t acc = 0;
  for (;;) {
    const IteratorStatus st = it->Read(it);
    if (st == ITERATOR_EOF) break;
    if (st != ITERATOR_OK) return -1;
    const RSIndexResult *r = it->current;
    acc += r->docId + r->freq;
    if (IndexResult_AggregateRef(r)) {
      const XVec *v = (const XVec *)r->data.agg.records;
      for (size_t c = 0; c < v->len; c++) acc += v->items[c]->freq + (uint64_t)v->items[c]->fieldMask;
    }
    n++;
  }
  if (checksum) *checksum = acc;
  return (long)n;
}
/* ops: 0 Read, 1 SkipTo(arg), 2 Rewind, 3 NumEstimated (status_out receives the estimate).  After every op: the
 * status, lastDocId, atEOF and the doc id of `current` (0 when NULL). */
XH_API void xh_iter_script(QueryIterator *it, size_t n_ops, const int *ops, const uint64_t *args, long *status_out,
                           uint64_t *last_out, uint8_t *eof_out, uint64_t *cur_out) {
  for (size_t i = 0; i < n_ops; i++) {
    switch (ops[i]) {
      case 0: status_out[i] = it->Read(it); break;
      case 1: status_out[i] = it->SkipTo(it, args[i]); break;
      case 2:
        it->Rewind(it);
        status_out[i] = 0;
        break;
      default: status_out[i] = (long)it->NumEstimated(it); break;
    }
    last_out[i] = it->lastDocId;
    eof_out[i] = it->atEOF;
    cur_out[i] = it->current ? it->current->docId : 0;
  }
}
/* rpscoreNext over the whole iterator: every result scored by `alias` with the document's metadata taken from arrays
 * indexed by doc id (entries beyond table_n: doc_len 0, score 1, max_freq 1).  Returns #hits or -1. */
XH_API long xh_iter_score_all(QueryIterator *it, const char *alias, const XScoreArgs *common, const uint32_t *doc_len,
                              const float *doc_score, const uint32_t *max_freq, size_t table_n, size_t cap, uint64_t *ids,
                              double *scores) {
  size_t n = 0;
  for (;;) {
    const IteratorStatus st = it->Read(it);
    if (st == ITERATOR_EOF) break;
    if (st != ITERATOR_OK) return -1;
    XScoreArgs a = *common;
    const uint64_t d = it->lastDocId;
    a.doc_len = d < table_n ? doc_len[d] : 0;
    a.doc_score = d < table_n ? doc_score[d] : 1.0f;
    a.max_term_freq = d < table_n ? max_freq[d] : 1;
    const double v = xh_score(alias, it->current, &a, NULL, 0);
    if (g_err[0]) return -1;
    if (n < cap) {
      ids[n] = d;
      scores[n] = v;
    }
    n++;
  }
  return (long)n;
}
XH_API void xh_iter_free(QueryIterator *it) {
  if (it) it->Free(it);
}
