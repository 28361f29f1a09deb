/*
 * oracle/scoring_oracle.c -- CPU restatement of RediSearch's built-in scorers and IDF.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/flat_oracle.c header for the rule).
 *
 * Follows, function by function:
 *   reference src/ext/default.c:68-106   tfidfRecursive
 *   reference src/ext/default.c:109-138  tfIdfInternal   (TFIDF / TFIDF.DOCNORM)
 *   reference src/ext/default.c:164-233  bm25Recursive / BM25Scorer (legacy BM25)
 *   reference src/ext/default.c:241-316  CalculateBM25Std / bm25StdRecursive / BM25StdScorer
 *   reference src/ext/default.c:329-359  tanhStretched / BM25StdTanhScorer
 *   reference src/ext/default.c:366-371  DocScoreScorer
 *   reference src/ext/default.c:378-461  dismaxRecursive / DisMaxScorer
 *   reference src/ext/default.c:475-497  HammingDistanceScorer
 *   reference src/redisearch_rs/idf/src/lib.rs:67-108  calculate_idf / calculate_idf_bm25
 *   reference src/index_result/index_result.c:51-103   IndexResult_MinOffsetDelta (slop)
 * The float/double promotion of the C source is kept literally: b and k1 are `float` constants,
 * `(float)doc_len`, everything else double.
 *
 * Pinned by tests/test_oracle_scoring.py against the reference's own known-answer tests
 * (idf/tests/tests.rs, tests/pytests/test_scorers.py, tests/pytests/test_vecsim.py:1248-1341).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* result-tree node kinds (same meaning as RSResultData_* in the reference) */
enum { R_UNION = 1, R_INTERSECTION = 2, R_TERM = 4, R_VIRTUAL = 8, R_NUMERIC = 16, R_METRIC = 32, R_HYBRID = 64 };

typedef struct ONode {
  int tag;
  double weight;
  uint32_t freq;
  int has_term;       /* Term nodes: whether an RSQueryTerm is attached */
  double idf;         /* QueryTerm_GetIDF */
  double bm25_idf;    /* QueryTerm_GetBM25_IDF */
  const uint32_t *offsets; /* decoded token positions (ascending) or NULL */
  size_t n_offsets;
  struct ONode **children;
  size_t n_children;
} ONode;

typedef struct {
  float score;          /* dmd->score */
  uint32_t max_term_freq;
  uint32_t doc_len;
} ODoc;

typedef struct {
  size_t num_docs;
  size_t num_terms;
  double avg_doc_len;
  uint64_t tanh_factor;
} OStats;

/* idf = logb(1 + (N+1)/max(n,1)) -- exponent extraction, exact */
double oracle_idf(size_t total_docs, size_t term_docs) {
  if (term_docs == 0) term_docs = 1;
  double value = 1.0 + (double)(total_docs + 1) / (double)term_docs;
  return (double)ilogb(value);
}
/* bm25_idf = ln(1 + (max(N,n) - n + 0.5)/(n + 0.5)) */
double oracle_idf_bm25(size_t total_docs, size_t term_docs) {
  if (total_docs < term_docs) total_docs = term_docs;
  double total = (double)total_docs, term = (double)term_docs;
  return log(1.0 + (total - term + 0.5) / (term + 0.5));
}

#define IS_AGG(t) ((t) & (R_INTERSECTION | R_UNION | R_HYBRID))

static int has_offsets(const ONode *r) {
  switch (r->tag) {
    case R_TERM: return r->n_offsets > 0;
    case R_INTERSECTION: case R_UNION: {
      int mask = 0;
      for (size_t i = 0; i < r->n_children; i++) mask |= r->children[i]->tag;
      return mask != R_VIRTUAL && mask != R_NUMERIC;
    }
    default: return 0;
  }
}
/* offsets iteration for slop: a Term child yields its positions; an aggregate child yields the ascending merge of its
 * descendants' positions (reference index_result/src/core/proximity.rs OffsetIter::Merge, pinned by the merge KATs at
 * :410-466 in tests/test_oracle_proximity.py through the byte-level restatement) */
#define OFF_EOF 0xFFFFFFFFu
typedef struct { uint32_t *pos; size_t n, i; int owned; } OffIt;
static size_t count_pos(const ONode *n) {
  if (n->tag == R_TERM) return n->n_offsets;
  size_t c = 0;
  if (IS_AGG(n->tag)) for (size_t k = 0; k < n->n_children; k++) c += count_pos(n->children[k]);
  return c;
}
static size_t gather_pos(const ONode *n, uint32_t *out) {
  if (n->tag == R_TERM) { if (n->n_offsets) memcpy(out, n->offsets, n->n_offsets * 4); return n->n_offsets; }
  size_t c = 0;
  if (IS_AGG(n->tag)) for (size_t k = 0; k < n->n_children; k++) c += gather_pos(n->children[k], out + c);
  return c;
}
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
static OffIt off_iter(const ONode *n) {
  OffIt it = {NULL, 0, 0, 0};
  if (n->tag == R_TERM) { it.pos = (uint32_t *)n->offsets; it.n = n->n_offsets; return it; }
  it.n = count_pos(n);
  it.pos = malloc((it.n ? it.n : 1) * 4);
  it.owned = 1;
  gather_pos(n, it.pos);
  qsort(it.pos, it.n, 4, cmp_u32);
  return it;
}
static void off_free(OffIt *it) { if (it->owned) free(it->pos); }
static uint32_t off_next(OffIt *it) { return it->i < it->n ? it->pos[it->i++] : OFF_EOF; }
#define ABSDELTA(x, y) ((x) > (y) ? (x) - (y) : (y) - (x))

int oracle_slop(const ONode *r) {
  if (!IS_AGG(r->tag)) return 1;
  size_t num = r->n_children;
  if (num <= 1) return 1;
  int dist = 0; size_t i = 0;
  while (i < num) {
    while (i < num && !has_offsets(r->children[i])) i++;
    if (i == num) break;
    OffIt v1 = off_iter(r->children[i]);
    i++;
    while (i < num && !has_offsets(r->children[i])) i++;
    if (i == num) { off_free(&v1); break; }
    OffIt v2 = off_iter(r->children[i]);
    uint32_t p1 = off_next(&v1), p2 = off_next(&v2);
    int cd = (int)ABSDELTA(p2, p1);
    while (cd > 1 && p1 != OFF_EOF && p2 != OFF_EOF) {
      int c = (int)ABSDELTA(p2, p1);
      if (c < cd) cd = c;
      if (p2 > p1) p1 = off_next(&v1); else p2 = off_next(&v2);
    }
    dist += cd * cd;
    off_free(&v1); off_free(&v2);
  }
  return dist ? (int)sqrt((double)dist) : (int)(num - 1);
}

/* ---- TFIDF ---------------------------------------------------------------------------------- */
static double tfidf_rec(const ONode *r) {
  if (r->tag == R_TERM) {
    double idf = r->has_term ? r->idf : 0;
    return r->weight * ((double)r->freq) * idf;
  }
  if (IS_AGG(r->tag)) {
    double ret = 0;
    for (size_t i = 0; i < r->n_children; i++) ret += tfidf_rec(r->children[i]);
    return r->weight * ret;
  }
  return r->weight * (double)r->freq;
}
/* norm_mode: 1 = max frequency (TFIDF), 2 = doc length (TFIDF.DOCNORM) */
double oracle_tfidf(const ONode *h, const ODoc *dmd, double min_score, int norm_mode) {
  if (dmd->score == 0) return 0;
  uint32_t norm = norm_mode == 1 ? dmd->max_term_freq : dmd->doc_len;
  if (norm == 0) return 0;
  double raw = tfidf_rec(h);
  double tfidf = dmd->score * raw / norm;
  if (tfidf < min_score) return 0;
  int slop = oracle_slop(h);
  tfidf /= slop;
  return tfidf;
}

/* ---- legacy BM25 ------------------------------------------------------------------------------ */
static double bm25_rec(const OStats *st, const ONode *r) {
  static const float b = 0.5;
  static const float k1 = 1.2;
  double f = (double)r->freq, ret = 0;
  if (r->tag == R_TERM) {
    double idf = r->has_term ? r->idf : 0;
    ret = r->weight * idf * f / (f + k1 * (1.0f - b + b * st->avg_doc_len));
  } else if (IS_AGG(r->tag)) {
    for (size_t i = 0; i < r->n_children; i++) ret += bm25_rec(st, r->children[i]);
    ret *= r->weight;
  } else if (f) {
    ret = r->weight * f / (f + k1 * (1.0f - b + b * st->avg_doc_len));
  }
  return ret;
}
double oracle_bm25(const OStats *st, const ONode *r, const ODoc *dmd, double min_score) {
  double res = bm25_rec(st, r);
  double score = dmd->score * res;
  if (score < min_score) return 0;
  int slop = oracle_slop(r);
  score /= slop;
  return score;
}

/* ---- BM25STD ---------------------------------------------------------------------------------- */
double oracle_bm25std_term(double idf, double f, int doc_len, double avg_doc_len, double weight) {
  const float b = 0.75f, k1 = 1.2f;
  return weight * idf * f * (k1 + 1) / (f + k1 * (1.0f - b + b * (float)doc_len / avg_doc_len));
}
static double bm25std_rec(const OStats *st, const ONode *r, const ODoc *dmd) {
  double f = (double)r->freq, ret = 0;
  if (r->tag == R_TERM) {
    ret = oracle_bm25std_term(r->bm25_idf, f, (int)dmd->doc_len, st->avg_doc_len, r->weight);
  } else if (IS_AGG(r->tag)) {
    for (size_t i = 0; i < r->n_children; i++) ret += bm25std_rec(st, r->children[i], dmd);
    ret *= r->weight;
  } else if (r->tag == R_VIRTUAL && f && r->weight) {
    ret = oracle_bm25std_term(1.0, 1, (int)dmd->doc_len, st->avg_doc_len, r->weight);
  }
  return ret;
}
double oracle_bm25std(const OStats *st, const ONode *r, const ODoc *dmd) {
  return dmd->score * bm25std_rec(st, r, dmd);
}
double oracle_bm25std_tanh(const OStats *st, const ONode *r, const ODoc *dmd) {
  double score = dmd->score * bm25std_rec(st, r, dmd);
  return tanh((1 / (double)st->tanh_factor) * score);
}

double oracle_docscore(const ODoc *dmd) { return dmd->score; }

/* ---- DISMAX ------------------------------------------------------------------------------------ */
double oracle_dismax(const ONode *r) {
  double ret = 0;
  switch (r->tag) {
    case R_TERM: case R_METRIC: case R_NUMERIC: case R_VIRTUAL:
      ret = r->freq; break;
    case R_INTERSECTION:
      for (size_t i = 0; i < r->n_children; i++) ret += oracle_dismax(r->children[i]);
      break;
    case R_UNION:
      for (size_t i = 0; i < r->n_children; i++) { double c = oracle_dismax(r->children[i]); if (c > ret) ret = c; }
      break;
    case R_HYBRID:
      return oracle_dismax(r->children[1]);
  }
  return r->weight * ret;
}

/* ---- HAMMING ----------------------------------------------------------------------------------- */
double oracle_hamming(const unsigned char *a, size_t alen, const unsigned char *b, size_t blen) {
  if (!blen || alen != blen) return 0;
  size_t ret = 0;
  for (size_t i = 0; i < alen; i++) ret += (size_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
  return 1.0 / (double)(ret + 1);
}

/* ---- flat (SoA) form of the scoring loop over an N-term intersection --------------------------
 * The shape bench/tests drive the GPU kernel with: hit m is
 * Intersection(weight=root_weight){ Term_t(weight[t], freq[t*M+m], idf[t]) }.
 * scorer: 0 BM25STD, 1 BM25STD.TANH, 2 BM25 (legacy, slop=1 as NOOFFSETS indexes have no
 * positions -> IndexResult_MinOffsetDelta returns num-1 for >1 offset-less children... see below),
 * 3 TFIDF, 4 TFIDF.DOCNORM, 5 DOCSCORE, 6 DISMAX.
 * For offset-less terms IndexResult_MinOffsetDelta yields dist==0 -> returns num-1 (reference
 * src/index_result/index_result.c:102); `slop_const` carries that value (1 for a single term). */
void oracle_score_flat(int scorer, size_t M, size_t T, const uint32_t *freq, const uint32_t *doc_len,
                       const uint32_t *max_freq, const float *doc_score, const double *idf,
                       const double *bm25_idf, const double *weight, double root_weight,
                       const OStats *st, double min_score, int slop_const, double *out) {
  enum { MAX_T = 64 };
  ONode terms[MAX_T]; ONode *kids[MAX_T]; ONode root;
  if (T > MAX_T) {  /* not a silent truncation: the caller sees NaN */
    for (size_t m = 0; m < M; m++) out[m] = NAN;
    return;
  }
  for (size_t m = 0; m < M; m++) {
    for (size_t t = 0; t < T; t++) {
      terms[t] = (ONode){R_TERM, weight[t], freq[t * M + m], 1, idf[t], bm25_idf[t], NULL, 0, NULL, 0};
      kids[t] = &terms[t];
    }
    root = (ONode){R_INTERSECTION, root_weight, 0, 0, 0, 0, NULL, 0, kids, T};
    for (size_t t = 0; t < T; t++) root.freq += terms[t].freq;
    ODoc d = {doc_score[m], max_freq ? max_freq[m] : 0, doc_len[m]};
    double s;
    switch (scorer) {
      case 0: s = oracle_bm25std(st, &root, &d); break;
      case 1: s = oracle_bm25std_tanh(st, &root, &d); break;
      case 2: { double res = bm25_rec(st, &root); s = d.score * res; if (s < min_score) s = 0; else s /= slop_const; break; }
      case 3: case 4: {
        if (d.score == 0) { s = 0; break; }
        uint32_t norm = scorer == 3 ? d.max_term_freq : d.doc_len;
        if (norm == 0) { s = 0; break; }
        s = d.score * tfidf_rec(&root) / norm;
        if (s < min_score) s = 0; else s /= slop_const;
        break;
      }
      case 5: s = d.score; break;
      default: s = oracle_dismax(&root); break;
    }
    out[m] = s;
  }
}

/* BM25STD.NORM: RPMaxScoreNormalizer (reference src/result_processor.c:1770-1812): accumulates every upstream result,
 * maxValue = MAX(maxValue, score) starting from 0, then yields score / maxValue -- unless maxValue == 0. */
void oracle_max_normalize(double *scores, size_t n) {
  double max_value = 0;
  for (size_t i = 0; i < n; i++) max_value = max_value > scores[i] ? max_value : scores[i];
  if (max_value != 0)
    for (size_t i = 0; i < n; i++) scores[i] = scores[i] / max_value;
}

/* ---- FT.HYBRID fusion (reference src/hybrid/hybrid_scoring.c:41-84, merger src/result_processor.c:2549-2571,
 * vector-score normalisation src/vector_normalization.h:37-60) -----------------------------------------------
 * Two ranked upstreams: a = search results (score descending), b = vector results (distance ascending, i.e.
 * normalised score descending).  At most `window` results are consumed from each, in upstream order; a
 * document found in both gets both contributions, added in upstream order (i = 0, then i = 1).
 *   RRF:    contribution = 1 / (constant + rank), rank = 1-based position in its upstream
 *   LINEAR: contribution = weight[i] * score;  the vector score is VectorNorm_<metric>(distance)
 * Output: every distinct document once, sorted by fused score descending, ties by lower doc id
 * (cmpByScore, src/result_processor.c:834-850).  metric < 0: b_scores are used as they are. */
double oracle_vector_norm(int metric, double d) {
  if (metric == 0) return 1.0 / (1.0 + d);          /* L2:     1/(1+distance)            */
  if (metric == 1) return (1.0 + d) / 2.0;           /* IP:     (1+dot)/2                 */
  if (metric == 2) return (1.0 + (1.0 - d)) / 2.0;   /* cosine: (1 + (1 - distance)) / 2  */
  return d;
}
typedef struct { uint64_t id; double s; } OFused;
static int cmp_fused(const void *x, const void *y) {
  const OFused *a = x, *b = y;
  if (a->s != b->s) return a->s > b->s ? -1 : 1;
  return a->id < b->id ? -1 : (a->id > b->id ? 1 : 0);
}
size_t oracle_hybrid_fuse(int scoring, double constant, double w0, double w1, int metric,
                          const uint64_t *a_ids, const double *a_scores, size_t na,
                          const uint64_t *b_ids, const double *b_scores, size_t nb, size_t window,
                          uint64_t *ids_out, double *scores_out) {
  if (na > window) na = window;
  if (nb > window) nb = window;
  OFused *f = malloc((na + nb + 1) * sizeof *f);
  size_t m = 0;
  for (size_t i = 0; i < na; i++) {
    double c = scoring == 0 ? 1.0 / (constant + (double)(i + 1)) : w0 * a_scores[i];
    f[m].id = a_ids[i]; f[m].s = 0.0 + c; m++;
  }
  for (size_t j = 0; j < nb; j++) {
    double c = scoring == 0 ? 1.0 / (constant + (double)(j + 1)) : w1 * oracle_vector_norm(metric, b_scores[j]);
    size_t hit = m;
    for (size_t i = 0; i < na; i++) if (f[i].id == b_ids[j]) { hit = i; break; }
    if (hit < m) f[hit].s += c;
    else { f[m].id = b_ids[j]; f[m].s = 0.0 + c; m++; }
  }
  qsort(f, m, sizeof *f, cmp_fused);
  for (size_t i = 0; i < m; i++) { ids_out[i] = f[i].id; scores_out[i] = f[i].s; }
  free(f);
  return m;
}
