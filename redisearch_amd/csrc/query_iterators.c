/*
 * query_iterators.c -- the reference's QueryIterator vtable (src/iterators/iterator_api.h:46-151) over MI355X hit
 * lists: librsgpu_iterators.so, Boundary 3 of SURVEY.md 8(b).  See include/rs_iterator.h.
 *
 * What replaces what:
 *   RSGPU_NewIntersectionIterator   NewIntersectionIterator (headers/iterators_ffi.h:309) over term children
 *                                   (NewInvIndIterator_TermQuery :404); behaviour of Intersection::read / skip_to /
 *                                   rewind / num_estimated, rqe_iterators/src/intersection.rs:428-530
 *   RSGPU_NewUnionIterator          NewUnionIterator over term children, union_flat.rs (full mode)
 *   RSGPU_NewNotIterator            NewNotIterator, not.rs:100-118,301 / not_optimized.rs
 * The AND / OR / NOT itself runs on the device when the iterator is created (rsgpu_search.h); this file is the host
 * side of the seam: it walks the hit list and rebuilds, per document, the RSIndexResult tree the reference's iterators
 * hold in `current` -- built with the module's own constructors (RSGPU_ResultAPI), so that the C pipeline and the
 * scorers cannot tell the difference.
 *
 * Doc ids of the whole hit list are mirrored on the host once (8 B per hit: SkipTo is a binary search); the per-term
 * records -- frequency, field mask, term-offset bytes -- are paged in blocks of `block` hits when a document of the
 * block is first positioned on.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rs_iterator.h"
#include "rsgpu_ext.h"

#define RSGPU_API __attribute__((visibility("default")))
#define MAX_CHILDREN 32
#define MAX_NODES 64

static __thread char g_err[256];
RSGPU_API const char *RSGPU_Iterators_LastError(void) { return g_err; }
static void set_err(const char *what, const char *detail) { snprintf(g_err, sizeof g_err, "%s%s%s", what, detail ? ": " : "", detail ? detail : ""); }

/* ---- the module's result constructors --------------------------------------------------------------------------------- */
static RSGPU_ResultAPI g_api;
static int g_api_set;
static size_t g_block = 65536;

RSGPU_API int RSGPU_Iterators_SetResultAPI(const RSGPU_ResultAPI *api, void *dl_handle) {
  if (api) {
    g_api = *api;
    g_api_set = 1;
    return 0;
  }
  void *h = dl_handle ? dl_handle : RTLD_DEFAULT;
  RSGPU_ResultAPI a;
  const char *missing = NULL;
#define BIND(name)                                   \
  do {                                               \
    *(void **)(&a.name) = dlsym(h, #name);           \
    if (!a.name && !missing) missing = #name;        \
  } while (0)
  BIND(NewIntersectResult);
  BIND(NewUnionResult);
  BIND(NewVirtualResult);
  BIND(NewTokenRecord);
  BIND(AggregateResult_AddChild);
  BIND(IndexResult_AggregateReset);
  BIND(IndexResult_Free);
  BIND(RSOffsetVector_SetData);
#undef BIND
  if (missing) {
    set_err("the RSIndexResult constructor is not visible in this process", missing);
    return -1;
  }
  g_api = a;
  g_api_set = 1;
  return 0;
}
RSGPU_API void RSGPU_Iterators_SetBlock(size_t hits) { g_block = hits ? hits : 65536; }

/* what a codec's records carry (include/rsgpu_search.h RSGPU_Codec; reference inverted_index/src/codec/) */
static int codec_has_freq(int c) {
  return c == RSGPU_CODEC_FULL || c == RSGPU_CODEC_FREQS_FIELDS || c == RSGPU_CODEC_FREQS_ONLY || c == RSGPU_CODEC_FREQS_OFFSETS ||
         c == RSGPU_CODEC_FULL_WIDE || c == RSGPU_CODEC_FREQS_FIELDS_WIDE;
}
static int codec_has_mask(int c) {
  return c == RSGPU_CODEC_FULL || c == RSGPU_CODEC_FREQS_FIELDS || c == RSGPU_CODEC_FIELDS_ONLY || c == RSGPU_CODEC_FIELDS_OFFSETS ||
         c >= RSGPU_CODEC_FULL_WIDE;
}

/* ---- the iterator ----------------------------------------------------------------------------------------------------- */
enum { K_AND = 0, K_OR = 1, K_NOT = 2 };
enum { G_TERM = 0, G_UNION = 1, G_INTERSECT = 2 }; /* RSGPU_OP_* of a root child */

typedef struct {
  int list;                /* index of the child's list in the hit list's lists */
  RSGPU_Postings *postings;
  RSIndexResult *rec;      /* Term record (NewTokenRecord) */
  int has_freq, has_mask;
  int offsets_set;         /* the record's offsets slice currently points into a block buffer */
  /* the current block's records */
  uint32_t *entry, *freq, *olen;
  uint64_t *mlo, *mhi, *opos;
  uint8_t *obytes;
  size_t obytes_cap;
  uint64_t obase;          /* byte position of obytes[0] in the list */
} Child;

typedef struct {
  QueryIterator base;
  int kind;
  RSGPU_Hits *hits;
  bool own_hits;
  size_t len, pos;         /* pos: index of the next hit Read() yields */
  size_t est;
  uint64_t *ids;           /* [len] */
  size_t n_children;
  Child child[MAX_CHILDREN];
  size_t block, blk_first, blk_count;  /* records of hits [blk_first, blk_first + blk_count) are loaded */
  RSIndexResult *result;   /* the aggregate / virtual result `current` points at */
  /* the result tree, post-order over the child slots (RSGPU_Hits_TreeNodes): node i is a term (child slot node_leaf[i]) or
   * an aggregate with its own record and the indices of its children; the root is the last node and owns `result` */
  int n_nodes;
  uint8_t node_op[MAX_NODES], node_leaf[MAX_NODES];
  uint8_t node_nkids[MAX_NODES];
  uint8_t node_kid[MAX_NODES][MAX_CHILDREN];
  RSIndexResult *node_rec[MAX_NODES]; /* aggregates: their record (the root's is `result`); terms: the child's record */
  uint8_t node_present[MAX_NODES];
  /* in-place rebuild: while the set of children present in a document equals the previous document's, every aggregate
   * already holds the right child pointers -- only the per-document fields are patched */
  bool wired;
  uint32_t last_pattern;
} GpuIt;

static size_t it_num_estimated(const QueryIterator *self) { return ((const GpuIt *)self)->est; }

static void free_child_block(Child *c) {
  free(c->entry), free(c->freq), free(c->olen), free(c->mlo), free(c->mhi), free(c->opos), free(c->obytes);
  c->entry = c->freq = c->olen = NULL;
  c->mlo = c->mhi = c->opos = NULL;
  c->obytes = NULL;
  c->obytes_cap = 0;
}

/* page in the records of the block that holds hit i */
static int load_block(GpuIt *it, size_t i) {
  const size_t first = i - i % it->block;
  size_t count = it->len - first;
  if (count > it->block) count = it->block;
  it->blk_count = 0; /* nothing is loaded until everything is */
  for (size_t c = 0; c < it->n_children; c++) {
    Child *ch = &it->child[c];
    if (!ch->entry) {
      ch->entry = malloc(it->block * sizeof *ch->entry);
      ch->freq = malloc(it->block * sizeof *ch->freq);
      ch->olen = malloc(it->block * sizeof *ch->olen);
      ch->mlo = malloc(it->block * sizeof *ch->mlo);
      ch->mhi = malloc(it->block * sizeof *ch->mhi);
      ch->opos = malloc(it->block * sizeof *ch->opos);
      if (!ch->entry || !ch->freq || !ch->olen || !ch->mlo || !ch->mhi || !ch->opos) {
        free_child_block(ch); /* all or nothing: the next attempt starts from scratch */
        set_err("out of memory", "a block of term records");
        return -1;
      }
    }
    if (RSGPU_Hits_ReadRecords(it->hits, (size_t)ch->list, first, count, ch->entry, ch->freq, ch->mlo, ch->mhi, ch->opos, ch->olen) !=
        (long)count) {
      set_err("RSGPU_Hits_ReadRecords", RSGPU_LastError());
      return -1;
    }
    /* the offsets blobs of the block lie in one byte range of the list (records are laid out in doc-id order) */
    uint64_t lo = ~0ull, hi = 0;
    for (size_t j = 0; j < count; j++)
      if (ch->entry[j] != 0xFFFFFFFFu && ch->olen[j]) {
        if (ch->opos[j] < lo) lo = ch->opos[j];
        if (ch->opos[j] + ch->olen[j] > hi) hi = ch->opos[j] + ch->olen[j];
      }
    ch->obase = 0;
    if (hi > lo) {
      if (hi - lo > ch->obytes_cap) {
        free(ch->obytes);
        ch->obytes_cap = (size_t)(hi - lo);
        ch->obytes = malloc(ch->obytes_cap);
        if (!ch->obytes) {
          ch->obytes_cap = 0;
          set_err("out of memory", "a block of term offsets");
          return -1;
        }
      }
      if (RSGPU_Postings_ReadBytes(ch->postings, (size_t)lo, (size_t)(hi - lo), ch->obytes)) {
        set_err("RSGPU_Postings_ReadBytes", RSGPU_LastError());
        return -1;
      }
      ch->obase = lo;
    }
  }
  it->blk_first = first;
  it->blk_count = count;
  return 0;
}

/* make sure the records of hit i are on the host; a failed device read surfaces as ITERATOR_TIMEOUT, the one
 * non-result status that leaves an iterator where it was (iterator_api.h:100-102) */
static int land(GpuIt *it, size_t i) {
  if (it->kind == K_NOT) return 0;
  if (i >= it->blk_first && i < it->blk_first + it->blk_count) return 0;
  return load_block(it, i);
}

/* Term record of a child slot for the hit at block position j (RawTermResultBuilder::new: frequency 1 unless the codec
 * decodes one; term.rs:95: every field unless it stores a mask).  The offsets go through the module's
 * RSOffsetVector_SetData (types_ffi.h:444; borrowed bytes of the block buffer) -- the slice's representation is the
 * module's business -- except that an empty slice is not set again over an empty slice (codecs without offsets). */
static inline void patch_term(Child *ch, size_t j, t_docId id) {
  RSIndexResult *t = ch->rec;
  t->docId = id;
  t->freq = ch->has_freq ? ch->freq[j] : 1;
  t->fieldMask = ch->has_mask ? ((t_fieldMask)ch->mhi[j] << 64) | (t_fieldMask)ch->mlo[j] : RS_FIELDMASK_ALL;
  const uint32_t ol = ch->olen[j];
  if (ol || ch->offsets_set) {
    g_api.RSOffsetVector_SetData(&t->data.term.offsets, ol ? (const char *)ch->obytes + (ch->opos[j] - ch->obase) : NULL, ol);
    ch->offsets_set = ol != 0;
  }
}

static void build_current(GpuIt *it, size_t i) {
  RSIndexResult *r = it->result;
  const t_docId id = it->ids[i];
  if (it->kind == K_NOT) {
    r->docId = id;
    it->base.lastDocId = id;
    it->base.current = r;
    return;
  }
  const size_t j = i - it->blk_first;
  /* which children hold this document (a union child -- or a whole nested aggregate -- may not) */
  uint32_t pattern = 0;
  for (size_t c = 0; c < it->n_children; c++) pattern |= (uint32_t)(it->child[c].entry[j] != 0xFFFFFFFFu) << c;
  if (it->wired && pattern == it->last_pattern) {
    /* same shape as the previous document: the aggregates keep their child pointers; what AggregateResult_AddChild
     * derives per child -- the doc id, the frequency sum, the field-mask union (intersection.rs:313-341) -- is recomputed */
    for (int n = 0; n < it->n_nodes; n++) {
      if (!it->node_present[n]) continue;
      if (it->node_op[n] == 0) {
        patch_term(&it->child[it->node_leaf[n]], j, id);
        continue;
      }
      RSIndexResult *a = it->node_rec[n];
      uint32_t f = 0;
      t_fieldMask m = 0;
      for (int k = 0; k < it->node_nkids[n]; k++) {
        const int kid = it->node_kid[n][k];
        if (!it->node_present[kid]) continue;
        f += it->node_rec[kid]->freq;
        m |= it->node_rec[kid]->fieldMask;
      }
      a->freq = f;
      a->fieldMask = m;
      a->docId = id;
    }
  } else {
    /* Intersection::build_aggregate_result (intersection.rs:313-341) / the union's (union_flat.rs:297-320), bottom-up:
     * per-document fields reset, then every child that holds the document pushed as a borrowed reference */
    for (int n = 0; n < it->n_nodes; n++) {
      if (it->node_op[n] == 0) {
        Child *ch = &it->child[it->node_leaf[n]];
        it->node_present[n] = ch->entry[j] != 0xFFFFFFFFu;
        if (it->node_present[n]) patch_term(ch, j, id);
        continue;
      }
      RSIndexResult *a = it->node_rec[n];
      a->freq = 0;
      a->fieldMask = 0;
      g_api.IndexResult_AggregateReset(a);
      int added = 0;
      for (int k = 0; k < it->node_nkids[n]; k++) {
        const int kid = it->node_kid[n][k];
        if (!it->node_present[kid]) continue;
        g_api.AggregateResult_AddChild(a, it->node_rec[kid]); /* takes the child's doc id, adds its frequency, ORs its mask */
        added++;
      }
      it->node_present[n] = added > 0;
    }
    r->docId = id;
    it->wired = true;
    it->last_pattern = pattern;
  }
  it->base.lastDocId = id;
  it->base.current = r;
}

static IteratorStatus it_eof(GpuIt *it) {
  it->base.atEOF = true;
  it->base.current = NULL; /* lastDocId stays on the last result yielded (iterator_api.h:96-99) */
  return ITERATOR_EOF;
}

static IteratorStatus it_read(QueryIterator *self) {
  GpuIt *it = (GpuIt *)self;
  if (it->base.atEOF || it->pos >= it->len) return it_eof(it);
  const size_t i = it->pos;
  if (land(it, i)) return ITERATOR_TIMEOUT;
  it->pos = i + 1;
  build_current(it, i);
  return ITERATOR_OK;
}

static IteratorStatus it_skip_to(QueryIterator *self, t_docId docId) {
  GpuIt *it = (GpuIt *)self;
  if (it->base.atEOF) return ITERATOR_EOF;
  /* first hit >= docId among those not yet passed */
  size_t lo = it->pos, hi = it->len;
  while (lo < hi) {
    const size_t mid = lo + (hi - lo) / 2;
    if (it->ids[mid] < docId) lo = mid + 1;
    else hi = mid;
  }
  if (lo >= it->len) {
    it->pos = it->len;
    return it_eof(it);
  }
  if (land(it, lo)) return ITERATOR_TIMEOUT;
  it->pos = lo + 1;
  build_current(it, lo);
  return it->ids[lo] == docId ? ITERATOR_OK : ITERATOR_NOTFOUND;
}

static void it_rewind(QueryIterator *self) {
  GpuIt *it = (GpuIt *)self;
  it->pos = 0;
  it->base.atEOF = false;
  it->base.lastDocId = 0;
  it->base.current = NULL;
}

/* The hit list is a snapshot of the lists as uploaded: nothing the index does afterwards moves it. */
static ValidateStatus it_revalidate(QueryIterator *self, struct IndexSpec *spec) {
  (void)self, (void)spec;
  return VALIDATE_OK;
}

static void it_free(QueryIterator *self) {
  if (!self) return;
  GpuIt *it = (GpuIt *)self;
  for (size_t c = 0; c < it->n_children; c++) {
    free_child_block(&it->child[c]);
    if (it->child[c].rec) {
      /* the record's offsets point into a block buffer that is gone now */
      g_api.RSOffsetVector_SetData(&it->child[c].rec->data.term.offsets, NULL, 0);
      g_api.IndexResult_Free(it->child[c].rec);
    }
  }
  /* aggregates: the children were borrowed (the root's record is `result`, the last node) */
  for (int n = 0; n < it->n_nodes; n++)
    if (it->node_op[n] != 0 && it->node_rec[n]) {
      g_api.IndexResult_AggregateReset(it->node_rec[n]);
      g_api.IndexResult_Free(it->node_rec[n]);
      if (it->node_rec[n] == it->result) it->result = NULL;
    }
  if (it->result) {
    if (it->kind != K_NOT) g_api.IndexResult_AggregateReset(it->result);
    g_api.IndexResult_Free(it->result);
  }
  if (it->own_hits && it->hits) RSGPU_Hits_Free(it->hits);
  free(it->ids);
  free(it);
}

/* Everything that can fail happens before the first record is created: the terms pass to the iterator if and only if
 * a non-NULL iterator is returned. */
static GpuIt *make(int kind, RSGPU_Hits *hits, bool own, const RSGPU_TermArg *terms, size_t num, double weight, size_t est) {
  if (!g_api_set && RSGPU_Iterators_SetResultAPI(NULL, NULL)) return NULL;
  int order[MAX_CHILDREN];
  if (kind != K_NOT) {
    const int nl = RSGPU_Hits_LeafOrder(hits, order);
    if (nl < 0 || (size_t)nl != num) {
      set_err("the hit list was not built from these terms", NULL);
      return NULL;
    }
  }
  GpuIt *it = calloc(1, sizeof *it);
  if (!it) return NULL;
  it->kind = kind;
  it->hits = hits;
  it->len = RSGPU_Hits_Len(hits);
  it->est = est;
  it->block = g_block;
  it->base.type = kind == K_AND ? IteratorType_Intersect : kind == K_OR ? IteratorType_Union : IteratorType_Not;
  it->base.NumEstimated = it_num_estimated;
  it->base.Read = it_read;
  it->base.SkipTo = it_skip_to;
  it->base.Revalidate = it_revalidate;
  it->base.Free = it_free;
  it->base.Rewind = it_rewind;
  it->base.ProfileChildren = NULL; /* a leaf as far as the profiler is concerned */
  it->base.PrintProfile = NULL;
  it->ids = malloc((it->len ? it->len : 1) * sizeof *it->ids);
  if (!it->ids || (it->len && RSGPU_Hits_ReadRange(hits, 0, it->len, it->ids) != (long)it->len)) {
    set_err("RSGPU_Hits_ReadRange", it->ids ? RSGPU_LastError() : "out of memory");
    free(it->ids);
    free(it);
    return NULL;
  }
  it->own_hits = own;
  if (kind == K_NOT) {
    it->result = g_api.NewVirtualResult(weight, RS_FIELDMASK_ALL); /* not.rs:112-115 */
    if (!it->result) {
      set_err("the module could not allocate a result", NULL);
      it->own_hits = false;
      it_free(&it->base);
      return NULL;
    }
    return it;
  }
  it->n_children = num;
  /* the result tree the device evaluated (any depth): terms = child slots, aggregates get records of their own; the root's
   * weight is this iterator's */
  int t_op[MAX_NODES], t_leaf[MAX_NODES], t_nch[MAX_NODES];
  double t_w[MAX_NODES];
  const int nn = RSGPU_Hits_TreeNodes(hits, t_op, t_leaf, t_nch, t_w);
  if (nn < 2 || nn > MAX_NODES || t_op[nn - 1] == 0) {
    set_err("the hit list carries no result tree", NULL);
    it->own_hits = false;
    it_free(&it->base);
    return NULL;
  }
  it->n_nodes = nn;
  {
    int size[MAX_NODES]; /* nodes in the subtree ending at i */
    for (int n = 0; n < nn; n++) {
      it->node_op[n] = (uint8_t)t_op[n];
      it->node_leaf[n] = (uint8_t)(t_op[n] == 0 ? t_leaf[n] : 0);
      it->node_nkids[n] = 0;
      size[n] = 1;
      if (t_op[n] == 0) continue;
      int at = n - 1, kids[MAX_CHILDREN], nk = 0;
      for (int k = 0; k < t_nch[n] && at >= 0 && nk < MAX_CHILDREN; k++) { /* the children, last first */
        kids[nk++] = at;
        size[n] += size[at];
        at -= size[at];
      }
      it->node_nkids[n] = (uint8_t)nk;
      for (int k = 0; k < nk; k++) it->node_kid[n][k] = (uint8_t)kids[nk - 1 - k];
    }
  }
  for (size_t c = 0; c < num; c++) { /* children in the order the device iterated them */
    Child *ch = &it->child[c];
    const RSGPU_TermArg *t = &terms[order[c]];
    ch->list = order[c];
    ch->postings = t->postings;
    const int codec = RSGPU_Postings_Codec(t->postings);
    ch->has_freq = codec_has_freq(codec);
    ch->has_mask = codec_has_mask(codec);
    ch->rec = g_api.NewTokenRecord(t->term, t->weight);
  }
  /* the module's allocator failing is the one thing that can go wrong after the terms have been handed over: the records
   * made so far (and their terms) are released with the iterator */
  bool ok = true;
  for (size_t c = 0; c < num; c++) ok = ok && it->child[c].rec != NULL;
  for (int n = 0; n < nn; n++) {
    if (it->node_op[n] == 0) {
      it->node_rec[n] = it->child[it->node_leaf[n]].rec;
      continue;
    }
    const double w = n == nn - 1 ? weight : t_w[n];
    it->node_rec[n] = it->node_op[n] == G_UNION ? g_api.NewUnionResult(it->node_nkids[n], w) : g_api.NewIntersectResult(it->node_nkids[n], w);
    ok = ok && it->node_rec[n] != NULL;
  }
  it->result = it->node_rec[nn - 1];
  if (!ok) {
    set_err("the module could not allocate a result", NULL);
    it->own_hits = false; /* the caller frees the hit list on a NULL return */
    it_free(&it->base);
    return NULL;
  }
  return it;
}

static int check_terms(const RSGPU_TermArg *terms, size_t num) {
  if (!terms || !num || num > MAX_CHILDREN) {
    set_err("1..32 terms", NULL);
    return -1;
  }
  for (size_t i = 0; i < num; i++)
    if (!terms[i].postings) {
      set_err("a term without a posting list", NULL);
      return -1;
    }
  return 0;
}

RSGPU_API QueryIterator *RSGPU_NewIntersectionIterator(const RSGPU_TermArg *terms, size_t num, int32_t max_slop, bool in_order,
                                                       double weight) {
  if (check_terms(terms, num)) return NULL;
  RSGPU_Postings *lists[MAX_CHILDREN];
  size_t est = (size_t)-1;
  for (size_t i = 0; i < num; i++) {
    lists[i] = terms[i].postings;
    /* num_expected = the smallest child estimate (intersection.rs:144-146); a term reader estimates its unique docs */
    const size_t n = RSGPU_Postings_NumEntries(lists[i]);
    if (n < est) est = n;
  }
  RSGPU_Hits *h = RSGPU_IntersectEx(lists, num, max_slop, in_order ? 1 : 0);
  if (!h) {
    set_err("RSGPU_IntersectEx", RSGPU_LastError());
    return NULL;
  }
  GpuIt *it = make(K_AND, h, true, terms, num, weight, est);
  if (!it) {
    RSGPU_Hits_Free(h);
    return NULL;
  }
  return &it->base;
}

RSGPU_API QueryIterator *RSGPU_NewUnionIterator(const RSGPU_TermArg *terms, size_t num, double weight) {
  if (check_terms(terms, num)) return NULL;
  RSGPU_Postings *lists[MAX_CHILDREN];
  size_t est = 0;
  for (size_t i = 0; i < num; i++) {
    lists[i] = terms[i].postings;
    est += RSGPU_Postings_NumEntries(lists[i]); /* union_flat.rs:102 */
  }
  RSGPU_Hits *h = RSGPU_Union(lists, num);
  if (!h) {
    set_err("RSGPU_Union", RSGPU_LastError());
    return NULL;
  }
  GpuIt *it = make(K_OR, h, true, terms, num, weight, est);
  if (!it) {
    RSGPU_Hits_Free(h);
    return NULL;
  }
  return &it->base;
}

RSGPU_API QueryIterator *RSGPU_NewNotIterator(RSGPU_Postings *child, RSGPU_Postings *universe, t_docId max_doc_id, double weight) {
  if (!child) {
    set_err("a NOT needs its child list", NULL);
    return NULL;
  }
  RSGPU_Hits *h = RSGPU_Not(child, universe, max_doc_id);
  if (!h) {
    set_err("RSGPU_Not", RSGPU_LastError());
    return NULL;
  }
  GpuIt *it = make(K_NOT, h, true, NULL, 0, weight, (size_t)max_doc_id /* not.rs:301-303 */);
  if (!it) {
    RSGPU_Hits_Free(h);
    return NULL;
  }
  return &it->base;
}

RSGPU_API QueryIterator *RSGPU_NewHitsIterator(RSGPU_Hits *hits, const RSGPU_TermArg *terms, size_t num, double weight, bool own_hits) {
  if (!hits || check_terms(terms, num)) return NULL;
  size_t mn = (size_t)-1, sum = 0;
  for (size_t i = 0; i < num; i++) {
    const size_t n = RSGPU_Postings_NumEntries(terms[i].postings);
    if (n < mn) mn = n;
    sum += n;
  }
  const int is_union = RSGPU_Hits_IsUnion(hits);
  GpuIt *it = make(is_union ? K_OR : K_AND, hits, own_hits, terms, num, weight, is_union ? sum : mn);
  return it ? &it->base : NULL;
}

RSGPU_API QueryIterator *RSGPU_NewTreeIterator(const RSGPU_TreeQuery *q, const RSGPU_TermArg *terms, double weight) {
  if (!q || !q->n_groups || !q->group_first || q->n_groups > MAX_CHILDREN) {
    set_err("an empty tree", NULL);
    return NULL;
  }
  const size_t num = q->group_first[q->n_groups];
  if (check_terms(terms, num)) return NULL;
  RSGPU_Postings *lists[MAX_CHILDREN];
  for (size_t i = 0; i < num; i++) lists[i] = terms[i].postings;
  RSGPU_TreeQuery tq = *q;
  tq.lists = lists;
  /* estimates as the reference's constructors compute them: a term its unique docs, a union the sum, an intersection
   * the smallest child (union_flat.rs:102, intersection.rs:144-146) */
  size_t est = q->root_op == RSGPU_OP_INTERSECT ? (size_t)-1 : 0;
  for (size_t g = 0; g < q->n_groups; g++) {
    const int op = q->group_op ? q->group_op[g] : RSGPU_OP_TERM;
    size_t e = op == RSGPU_OP_INTERSECT ? (size_t)-1 : 0;
    for (size_t l = q->group_first[g]; l < q->group_first[g + 1]; l++) {
      const size_t n = RSGPU_Postings_NumEntries(lists[l]);
      if (op == RSGPU_OP_INTERSECT) e = n < e ? n : e;
      else e += n;
    }
    if (q->root_op == RSGPU_OP_INTERSECT) est = e < est ? e : est;
    else est += e;
  }
  RSGPU_Hits *h = RSGPU_EvalTree(&tq);
  if (!h) {
    set_err("RSGPU_EvalTree", RSGPU_LastError());
    return NULL;
  }
  GpuIt *it = make(RSGPU_Hits_IsUnion(h) ? K_OR : K_AND, h, true, terms, num, weight, est);
  if (!it) {
    RSGPU_Hits_Free(h);
    return NULL;
  }
  return &it->base;
}

/* A query tree of any depth behind one iterator (RSGPU_EvalTreeNodes): `current` nests aggregates exactly as the
 * reference's nested iterators do. */
RSGPU_API QueryIterator *RSGPU_NewTreeNodesIterator(const RSGPU_TreeNode *nodes, size_t n_nodes, const RSGPU_TermArg *terms,
                                                    size_t num, double weight) {
  if (!nodes || !n_nodes || n_nodes > MAX_NODES || check_terms(terms, num)) {
    if (!nodes || !n_nodes || n_nodes > MAX_NODES) set_err("1..64 tree nodes", NULL);
    return NULL;
  }
  RSGPU_Postings *lists[MAX_CHILDREN];
  for (size_t i = 0; i < num; i++) lists[i] = terms[i].postings;
  /* estimates as the reference's constructors compute them, bottom-up: a term its unique docs, a union the sum, an
   * intersection the smallest child (union_flat.rs:102, intersection.rs:144-146) */
  size_t stack[MAX_NODES], sp = 0;
  for (size_t i = 0; i < n_nodes; i++) {
    if (nodes[i].op == RSGPU_OP_TERM) {
      if (nodes[i].list >= num) {
        set_err("a term node names a list that is not there", NULL);
        return NULL;
      }
      stack[sp++] = RSGPU_Postings_NumEntries(lists[nodes[i].list]);
      continue;
    }
    if (!nodes[i].n_children || nodes[i].n_children > sp) {
      set_err("the node array is not a post-order tree", NULL);
      return NULL;
    }
    size_t e = nodes[i].op == RSGPU_OP_INTERSECT ? (size_t)-1 : 0;
    for (size_t k = 0; k < nodes[i].n_children; k++) {
      const size_t c = stack[--sp];
      if (nodes[i].op == RSGPU_OP_INTERSECT) e = c < e ? c : e;
      else e += c;
    }
    stack[sp++] = e;
  }
  if (sp != 1) {
    set_err("the node array is not a post-order tree", NULL);
    return NULL;
  }
  RSGPU_Hits *h = RSGPU_EvalTreeNodes(nodes, n_nodes, lists, num);
  if (!h) {
    set_err("RSGPU_EvalTreeNodes", RSGPU_LastError());
    return NULL;
  }
  GpuIt *it = make(RSGPU_Hits_IsUnion(h) ? K_OR : K_AND, h, true, terms, num, weight, stack[0]);
  if (!it) {
    RSGPU_Hits_Free(h);
    return NULL;
  }
  return &it->base;
}

RSGPU_API RSGPU_Hits *RSGPU_Iterator_Hits(QueryIterator *it) {
  if (!it || it->Read != it_read) return NULL;
  return ((GpuIt *)it)->hits;
}
