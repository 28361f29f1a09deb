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
  /* two-level trees (RSGPU_NewTreeIterator): the root's children are groups of consecutive child slots; a group that is
   * a union / intersection has its own aggregate between the root and its term records */
  int n_groups;            /* 0: flat (every child hangs off the root) */
  int group_first[MAX_CHILDREN + 1], group_op[MAX_CHILDREN];
  RSIndexResult *group_rec[MAX_CHILDREN];
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

static void build_current(GpuIt *it, size_t i) {
  RSIndexResult *r = it->result;
  const t_docId id = it->ids[i];
  if (it->kind == K_NOT) {
    r->docId = id;
  } else {
    /* Intersection::build_aggregate_result (intersection.rs:313-341): per-document fields reset, then every child
     * pushed as a borrowed reference -- AddChild takes the child's doc id, adds its frequency, ORs its field mask */
    r->freq = 0;
    r->fieldMask = 0;
    g_api.IndexResult_AggregateReset(r);
    const size_t j = i - it->blk_first;
    const int ng = it->n_groups ? it->n_groups : (int)it->n_children;
    for (int g = 0; g < ng; g++) {
      const size_t c0 = it->n_groups ? (size_t)it->group_first[g] : (size_t)g;
      const size_t c1 = it->n_groups ? (size_t)it->group_first[g + 1] : (size_t)g + 1;
      RSIndexResult *parent = r;
      if (it->n_groups && it->group_op[g] != G_TERM) { /* a nested Union / Intersection: rebuilt like the root */
        parent = it->group_rec[g];
        parent->freq = 0;
        parent->fieldMask = 0;
        g_api.IndexResult_AggregateReset(parent);
      }
      size_t added = 0;
      for (size_t c = c0; c < c1; c++) {
        Child *ch = &it->child[c];
        if (ch->entry[j] == 0xFFFFFFFFu) continue; /* a union child (or a whole group) that does not hold the document */
        RSIndexResult *t = ch->rec;
        t->docId = id;
        t->freq = ch->has_freq ? ch->freq[j] : 1; /* RawTermResultBuilder::new: frequency 1 unless the codec decodes one */
        t->fieldMask = ch->has_mask ? ((t_fieldMask)ch->mhi[j] << 64) | (t_fieldMask)ch->mlo[j] : RS_FIELDMASK_ALL; /* term.rs:95 */
        const char *ob = ch->olen[j] ? (const char *)ch->obytes + (ch->opos[j] - ch->obase) : NULL;
        g_api.RSOffsetVector_SetData(&t->data.term.offsets, ob, ch->olen[j]);
        g_api.AggregateResult_AddChild(parent, t);
        added++;
      }
      if (parent != r && added) g_api.AggregateResult_AddChild(r, parent); /* takes the group's doc id, frequency, mask */
    }
    r->docId = id;
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
  if (it->result) {
    if (it->kind != K_NOT) g_api.IndexResult_AggregateReset(it->result); /* the children were borrowed */
    g_api.IndexResult_Free(it->result);
  }
  for (int g = 0; g < it->n_groups; g++)
    if (it->group_rec[g]) {
      g_api.IndexResult_AggregateReset(it->group_rec[g]);
      g_api.IndexResult_Free(it->group_rec[g]);
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
  it->result = kind == K_AND ? g_api.NewIntersectResult(num, weight) : g_api.NewUnionResult(num, weight);
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
  bool ok = it->result != NULL;
  for (size_t c = 0; c < num; c++) ok = ok && it->child[c].rec != NULL;
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
  int gop[MAX_CHILDREN];
  const int ng = RSGPU_Hits_Tree(hits, NULL, NULL, gop, NULL);
  for (int g = 0; g < ng; g++)
    if (gop[g] != G_TERM) { /* nested groups need the tree's shape and weights: RSGPU_NewTreeIterator builds those */
      set_err("the hit list comes from a two-level tree", "use RSGPU_NewTreeIterator");
      return NULL;
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
  int is_union = 0, gf[MAX_CHILDREN + 1], gop[MAX_CHILDREN];
  double gw[MAX_CHILDREN];
  const int ng = RSGPU_Hits_Tree(h, &is_union, gf, gop, gw);
  GpuIt *it = ng > 0 ? make(is_union ? K_OR : K_AND, h, true, terms, num, weight, est) : NULL;
  if (!it) {
    RSGPU_Hits_Free(h);
    return NULL;
  }
  /* make() hung every term off the root; re-root: the root holds one child per group */
  g_api.IndexResult_AggregateReset(it->result);
  g_api.IndexResult_Free(it->result);
  it->result = is_union ? g_api.NewUnionResult((size_t)ng, weight) : g_api.NewIntersectResult((size_t)ng, weight);
  it->n_groups = ng;
  for (int g = 0; g <= ng; g++) it->group_first[g] = gf[g];
  for (int g = 0; g < ng; g++) {
    it->group_op[g] = gop[g];
    const size_t n = (size_t)(gf[g + 1] - gf[g]);
    it->group_rec[g] = gop[g] == G_UNION ? g_api.NewUnionResult(n, gw[g]) : gop[g] == G_INTERSECT ? g_api.NewIntersectResult(n, gw[g]) : NULL;
  }
  bool ok = it->result != NULL;
  for (int g = 0; g < ng; g++) ok = ok && (gop[g] == G_TERM || it->group_rec[g] != NULL);
  if (!ok) { /* the module's allocator failed (see make()): everything made so far goes with the iterator */
    set_err("the module could not allocate a result", NULL);
    it_free(&it->base); /* (owns the hit list) */
    return NULL;
  }
  return &it->base;
}

RSGPU_API RSGPU_Hits *RSGPU_Iterator_Hits(QueryIterator *it) {
  if (!it || it->Read != it_read) return NULL;
  return ((GpuIt *)it)->hits;
}
