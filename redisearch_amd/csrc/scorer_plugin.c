/*
 * scorer_plugin.c -> librsgpu_scorers.so: the per-result side of Boundary 2.
 *
 * RediSearch's result pipeline scores ONE result per call (rpscoreNext, reference src/result_processor.c:570-603)
 * through an RSScoringFunction it found by alias (src/extension.c:151-166).  That granularity cannot feed a GPU, so
 * the batch entry point RSGPU_Hits_Score (include/rsgpu_search.h) stays the fast path; this plugin is what makes
 * `SCORER BM25STD` & co. reachable at all for a module that was not rebuilt around the batch ABI: a plain-C shared
 * object the module loads with `EXTLOAD`, exporting RS_ExtensionInit, computing on the host exactly what the device
 * score_kernel computes (fp64, the C source's float constants, same operation order), for result trees of any
 * depth, with EXPLAINSCORE.
 *
 * Shape (deliberately not the reference's four hand-unrolled recursions, src/ext/default.c:68-461): ONE tree
 * walk parameterised by a scorer family; a family supplies the leaf value and says how an aggregate folds its
 * children (sum, or max for a DISMAX union).  Scorers differ only in the epilogue applied to the root value.
 *
 * Undefined on purpose (bound to the module's Rust accessors at load time, include/rs_extension.h):
 * IndexResult_QueryTermRef, IndexResult_AggregateRefUnchecked, AggregateResult_GetRecordsSlice,
 * AggregateResult_Get, QueryTerm_GetIDF, QueryTerm_GetBM25_IDF, QueryTerm_GetStrAndLen.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rs_extension.h"

/* ---- allocator for EXPLAINSCORE nodes -------------------------------------------------------------------------- */
static void *(*g_calloc)(size_t, size_t);
static void (*g_free)(void *);

__attribute__((visibility("default"))) void RSGPU_Scorers_SetAllocator(void *(*calloc_fn)(size_t, size_t),
                                                                       void (*free_fn)(void *)) {
  g_calloc = calloc_fn;
  g_free = free_fn;
}

static void bind_allocator(void) {
  if (g_calloc && g_free) return;
  /* the module's redismodule.h globals are pointers to function pointers */
  void *(**mc)(size_t, size_t) = (void *(**)(size_t, size_t))dlsym(RTLD_DEFAULT, "RedisModule_Calloc");
  void (**mf)(void *) = (void (**)(void *))dlsym(RTLD_DEFAULT, "RedisModule_Free");
  if (mc && *mc && mf && *mf) {
    g_calloc = *mc;
    g_free = *mf;
  } else {
    g_calloc = calloc;
    g_free = free;
  }
}

/* Replace node->str by the formatted text (what the module's explain() does, src/score_explain.c:54-63). */
__attribute__((format(printf, 2, 3))) static void say(RSScoreExplain *node, const char *fmt, ...) {
  if (!node) return;
  char stack[512];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(stack, sizeof stack, fmt, ap);
  va_end(ap);
  if (n < 0) return;
  char *s = (char *)g_calloc((size_t)n + 1, 1);
  if (!s) return;
  if ((size_t)n < sizeof stack) {
    memcpy(s, stack, (size_t)n + 1);
  } else {
    va_start(ap, fmt);
    vsnprintf(s, (size_t)n + 1, fmt, ap);
    va_end(ap);
  }
  if (node->str) g_free(node->str);
  node->str = s;
}

/* The explanation so far becomes the only child of a fresh root, and the caller sees the new root through
 * args->scrExp (src/ext/default.c:58-65: rpscoreNext reads the pointer back after the call). */
static RSScoreExplain *push_root(const ScoringFunctionArgs *args, RSScoreExplain *cur) {
  if (!cur) return NULL;
  RSScoreExplain *root = (RSScoreExplain *)g_calloc(1, sizeof *root);
  if (!root) return cur;
  root->numChildren = 1;
  root->children = cur;
  ((ScoringFunctionArgs *)args)->scrExp = root;
  return root;
}

/* ---- the tree walk ------------------------------------------------------------------------------------------------- */
typedef enum { FAM_TFIDF, FAM_BM25, FAM_BM25STD, FAM_DISMAX } family_t;

typedef struct {
  family_t fam;
  double avg_doc_len;
  int doc_len;
} walk_t;

/* constants as the C source spells them: float, so that `1.0f - b` and `k1 + 1` round in float first */
static const float kLegacyB = 0.5f, kStdB = 0.75f, kK1 = 1.2f;

static inline double idf_of(const RSQueryTerm *t) { return t ? QueryTerm_GetIDF(t) : 0.0; }

static inline double bm25std_term(double idf, double f, int doc_len, double avg, double weight) {
  /* weight*idf*f*(k1+1) / (f + k1*(1 - b + b*len/avg)), src/ext/default.c:250-258 */
  const double len_ratio = (double)(kStdB * (float)doc_len) / avg;
  const double denom = f + (double)kK1 * ((double)(1.0f - kStdB) + len_ratio);
  return weight * idf * f * (double)(kK1 + 1) / denom;
}

static double leaf(const walk_t *w, const RSIndexResult *r, RSScoreExplain *e) {
  const uint8_t tag = r->data.tag;
  const double f = (double)r->freq;
  switch (w->fam) {
    case FAM_TFIDF: {
      if (tag == RSResultData_Term) {
        const double idf = idf_of(IndexResult_QueryTermRef(r));
        const double v = r->weight * f * idf;
        say(e, "(TFIDF %.2f = Weight %.2f * TF %d * IDF %.2f)", v, r->weight, (int)r->freq, idf);
        return v;
      }
      say(e, "(TFIDF %.2f = Weight %.2f * Frequency %d)", r->weight * f, r->weight, (int)r->freq);
      return r->weight * f;
    }
    case FAM_BM25: {
      /* legacy: k1*(1 - b + b*avgDocLen) -- the average, not the document's ratio (src/ext/default.c:166-209) */
      const double sat = (double)kK1 * ((double)(1.0f - kLegacyB) + (double)kLegacyB * w->avg_doc_len);
      if (tag == RSResultData_Term) {
        const double idf = idf_of(IndexResult_QueryTermRef(r));
        const double v = r->weight * idf * f / (f + sat);
        say(e, "(%.2f = Weight %.2f * IDF %.2f * F %d / (F %d + k1 1.2 * (1 - b 0.5 + b 0.5 * Average Len %.2f)))", v,
            r->weight, idf, (int)r->freq, (int)r->freq, w->avg_doc_len);
        return v;
      }
      if (r->freq) {
        const double v = r->weight * f / (f + sat);
        say(e, "(%.2f = Weight %.2f * F %d / (F %d + k1 1.2 * (1 - b 0.5 + b 0.5 * Average Len %.2f)))", v, r->weight,
            (int)r->freq, (int)r->freq, w->avg_doc_len);
        return v;
      }
      say(e, "Frequency 0 -> value 0");
      return 0.0;
    }
    case FAM_BM25STD: {
      static const char kFmt[] =
          "%.*s: (%.2f = Weight %.2f * IDF %.2f * (F %.2f * (k1 1.2 + 1)) / (F %.2f + k1 1.2 * (1 - b 0.75 + b 0.75 *"
          " Doc Len %d / Average Doc Len %.2f)))";
      if (tag == RSResultData_Term) {
        const RSQueryTerm *t = IndexResult_QueryTermRef(r);
        const double idf = t ? QueryTerm_GetBM25_IDF(t) : 0.0;
        const double v = bm25std_term(idf, f, w->doc_len, w->avg_doc_len, r->weight);
        if (e) {
          size_t n = 0;
          const char *s = t ? QueryTerm_GetStrAndLen(t, &n) : "";
          say(e, kFmt, (int)n, s ? s : "", v, r->weight, idf, f, f, w->doc_len, w->avg_doc_len);
        }
        return v;
      }
      if (tag == RSResultData_Virtual && r->freq && r->weight != 0.0) {
        /* wildcard: only the weight and the document's length count (idf = f = 1), src/ext/default.c:296-301 */
        const double v = bm25std_term(1.0, 1.0, w->doc_len, w->avg_doc_len, r->weight);
        say(e, kFmt, 1, "*", v, r->weight, 1.0, 1.0, 1.0, w->doc_len, w->avg_doc_len);
        return v;
      }
      say(e, "Irrelevant token -> score is 0");
      return 0.0;
    }
    case FAM_DISMAX: {
      say(e, "DISMAX %.2f = Weight %.2f * Frequency %d", r->weight * f, r->weight, (int)r->freq);
      return r->weight * f;
    }
  }
  return 0.0;
}

static double walk(const walk_t *w, const RSIndexResult *r, RSScoreExplain *e) {
  const uint8_t tag = r->data.tag;
  const uint8_t aggregates = w->fam == FAM_DISMAX
                                 ? (RSResultData_Intersection | RSResultData_Union)
                                 : (RSResultData_Intersection | RSResultData_Union | RSResultData_HybridMetric);
  if (w->fam == FAM_DISMAX && tag == RSResultData_HybridMetric) {
    /* the text child of a hybrid (vector, text) pair, unweighted (src/ext/default.c:442-448) */
    return walk(w, AggregateResult_Get(IndexResult_AggregateRefUnchecked(r), 1), e);
  }
  if (!(tag & aggregates)) return leaf(w, r, e);

  const AggregateRecordsSlice kids = AggregateResult_GetRecordsSlice(IndexResult_AggregateRefUnchecked(r));
  if (e) {
    e->numChildren = (int)kids.len;
    e->children = (RSScoreExplain *)g_calloc(kids.len ? kids.len : 1, sizeof(RSScoreExplain));
    if (!e->children) e->numChildren = 0;
  }
  const int take_max = w->fam == FAM_DISMAX && tag == RSResultData_Union;
  double acc = 0.0;
  for (size_t i = 0; i < kids.len; i++) {
    const double v = walk(w, kids.ptr[i], e && e->children ? &e->children[i] : NULL);
    acc = take_max ? (acc > v ? acc : v) : acc + v;
  }
  switch (w->fam) {
    case FAM_TFIDF: say(e, "(Weight %.2f * total children TFIDF %.2f)", r->weight, acc); break;
    case FAM_DISMAX: say(e, "%.2f = Weight %.2f * children DISMAX %.2f", r->weight * acc, r->weight, acc); break;
    default: say(e, "(Weight %.2f * children BM25 %.2f)", r->weight, acc); break;
  }
  return r->weight * acc;
}

/* ---- the nine scorers ------------------------------------------------------------------------------------------------- */
static RSScoreExplain *begin(const ScoringFunctionArgs *args) {
  if (args->scrExp) bind_allocator();
  return (RSScoreExplain *)args->scrExp;
}

static double tfidf_common(const ScoringFunctionArgs *args, const RSIndexResult *res, const RSDocumentMetadata *dmd,
                           double min_score, int by_doc_len) {
  RSScoreExplain *e = begin(args);
  if (dmd->score == 0) {
    say(e, "Document score is 0");
    return 0;
  }
  const uint32_t norm = by_doc_len ? dmd->docLen : dmd->maxTermFreq;
  if (norm == 0) {
    say(e, "Document %s is 0", by_doc_len ? "length" : "max frequency");
    return 0;
  }
  const walk_t w = {FAM_TFIDF, args->indexStats.avgDocLen, (int)dmd->docLen};
  const double raw = walk(&w, res, e);
  double v = (double)dmd->score * raw / (double)norm;
  e = push_root(args, e);
  if (v < min_score) { /* the slop would only lower it further */
    say(e, "TFIDF score of %.2f is smaller than minimum score %.2f", v, min_score);
    return 0;
  }
  const int slop = args->GetSlop(res);
  v /= slop;
  say(e, "Final TFIDF : words TFIDF %.2f * document score %.2f / norm %d / slop %d", raw, dmd->score, (int)norm, slop);
  return v;
}

static double score_tfidf(const ScoringFunctionArgs *a, const RSIndexResult *r, const RSDocumentMetadata *d, double m) {
  return tfidf_common(a, r, d, m, 0);
}
static double score_tfidf_docnorm(const ScoringFunctionArgs *a, const RSIndexResult *r, const RSDocumentMetadata *d,
                                  double m) {
  return tfidf_common(a, r, d, m, 1);
}

static double score_bm25(const ScoringFunctionArgs *args, const RSIndexResult *res, const RSDocumentMetadata *dmd,
                         double min_score) {
  RSScoreExplain *e = begin(args);
  const walk_t w = {FAM_BM25, args->indexStats.avgDocLen, (int)dmd->docLen};
  const double words = walk(&w, res, e);
  double v = (double)dmd->score * words;
  e = push_root(args, e);
  if (v < min_score) {
    /* argument order as the reference prints it (src/ext/default.c:222) */
    say(e, "BM25 score of %.2f is smaller than minimum score %.2f", words, v);
    return 0;
  }
  const int slop = args->GetSlop(res);
  v /= slop;
  say(e, "Final BM25 : words BM25 %.2f * document score %.2f / slop %d", words, dmd->score, slop);
  return v;
}

static double bm25std_root(const ScoringFunctionArgs *args, const RSIndexResult *res, const RSDocumentMetadata *dmd,
                           RSScoreExplain **e_io) {
  const walk_t w = {FAM_BM25STD, args->indexStats.avgDocLen, (int)dmd->docLen};
  const double words = walk(&w, res, *e_io);
  const double v = (double)dmd->score * words;
  *e_io = push_root(args, *e_io);
  say(*e_io, "Final BM25 : words BM25 %.2f * document score %.2f", words, dmd->score);
  return v;
}

/* Also registered as BM25STD.NORM: the division by the maximum is a result processor behind the scorer
 * (RPMaxScoreNormalizer, src/result_processor.c:1770-1812), not part of the scoring function. */
static double score_bm25std(const ScoringFunctionArgs *args, const RSIndexResult *res, const RSDocumentMetadata *dmd,
                            double min_score) {
  (void)min_score;
  RSScoreExplain *e = begin(args);
  return bm25std_root(args, res, dmd, &e);
}

static double score_bm25std_tanh(const ScoringFunctionArgs *args, const RSIndexResult *res,
                                 const RSDocumentMetadata *dmd, double min_score) {
  (void)min_score;
  RSScoreExplain *e = begin(args);
  const double v = bm25std_root(args, res, dmd, &e);
  const double out = tanh((1 / (double)args->tanhFactor) * v);
  e = push_root(args, e);
  say(e, "Final Normalized BM25 : tanh(stretch factor 1/%d * Final BM25 %.2f)", (int)args->tanhFactor, v);
  return out;
}

static double score_docscore(const ScoringFunctionArgs *args, const RSIndexResult *res, const RSDocumentMetadata *dmd,
                             double min_score) {
  (void)res;
  (void)min_score;
  say(begin(args), "Document's score is %.2f", dmd->score);
  return dmd->score;
}

static double score_dismax(const ScoringFunctionArgs *args, const RSIndexResult *res, const RSDocumentMetadata *dmd,
                           double min_score) {
  (void)min_score;
  const walk_t w = {FAM_DISMAX, args->indexStats.avgDocLen, (int)dmd->docLen};
  return walk(&w, res, begin(args));
}

/* 1 / (1 + popcount(query payload XOR document payload)); 0 unless both payloads have the same non-zero length
 * (src/ext/default.c:475-499). */
static double score_hamming(const ScoringFunctionArgs *args, const RSIndexResult *res, const RSDocumentMetadata *dmd,
                            double min_score) {
  (void)res;
  (void)min_score;
  RSScoreExplain *e = begin(args);
  if (!(dmd->flags & RS_DOCUMENT_HAS_PAYLOAD) || !dmd->payload || !dmd->payload->len ||
      dmd->payload->len != args->qdatalen) {
    say(e, "Payloads provided to scorer vary in length");
    return 0;
  }
  const size_t len = args->qdatalen;
  const unsigned char *a = (const unsigned char *)args->qdata, *b = (const unsigned char *)dmd->payload->data;
  size_t bits = 0, i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t x, y;
    memcpy(&x, a + i, 8);
    memcpy(&y, b + i, 8);
    bits += (size_t)__builtin_popcountll(x ^ y);
  }
  for (; i < len; i++) bits += (size_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
  const double v = 1.0 / (double)(bits + 1);
  say(e, "String length is %zu. Bit count is %zu. Result is (1 / count + 1) = %.2f", len, bits, v);
  return v;
}

/* ---- registration ---------------------------------------------------------------------------------------------------- */
static const struct {
  const char *alias;
  RSScoringFunction fn;
} kScorers[] = {
    {TFIDF_SCORER_NAME, score_tfidf},
    {DISMAX_SCORER_NAME, score_dismax},
    {BM25_SCORER_NAME, score_bm25},
    {BM25_STD_SCORER_NAME, score_bm25std},
    {BM25_STD_NORMALIZED_TANH_SCORER_NAME, score_bm25std_tanh},
    {BM25_STD_NORMALIZED_MAX_SCORER_NAME, score_bm25std},
    {HAMMINGDISTANCE_SCORER, score_hamming},
    {TFIDF_DOCNORM_SCORER_NAME, score_tfidf_docnorm},
    {DOCSCORE_SCORER, score_docscore},
};

__attribute__((visibility("default"))) int RS_ExtensionInit(RSExtensionCtx *ctx) {
  if (!ctx || !ctx->RegisterScoringFunction) return REDISEARCH_ERR;
  for (size_t i = 0; i < sizeof kScorers / sizeof kScorers[0]; i++) {
    if (ctx->RegisterScoringFunction(kScorers[i].alias, kScorers[i].fn, NULL, NULL) == REDISEARCH_OK) continue;
    char prefixed[64];
    snprintf(prefixed, sizeof prefixed, "%s%s", RSGPU_SCORER_ALIAS_PREFIX, kScorers[i].alias);
    /* the registry keeps its own copy of the alias (TrieMap_Add, src/extension.c:84) */
    if (ctx->RegisterScoringFunction(prefixed, kScorers[i].fn, NULL, NULL) != REDISEARCH_OK) return REDISEARCH_ERR;
  }
  return REDISEARCH_OK;
}
