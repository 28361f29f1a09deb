/*
 * oracle/flat_oracle.c -- CPU restatement of the VecSim FLAT (brute-force) index.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under redisearch_amd/ may call, link or import this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * Provenance.  The arithmetic of this path lives in the third-party dependency
 * RedisAI/VectorSimilarity (reference .gitmodules:10-12, path deps/VectorSimilarity), which is an
 * EMPTY un-vendored submodule in /root/reference; its pinned commit is unrecoverable (no .git).
 * This file therefore restates the library's published brute-force algorithm and anchors every
 * behaviour on the reference's own call sites and tests:
 *   - distances: L2 = sum (a-b)^2 (squared), IP = 1 - dot, COSINE = 1 - dot of normalised
 *     vectors (reference tests/pytests/test_vecsim.py:87-90,165-168;
 *     tests/pytests/test_hybrid_vector_normalizer.py:57-58);
 *   - replies sorted ascending by score (reference
 *     src/redisearch_rs/vector_score_source/tests/integration/source_pytest_parity.rs:35-46);
 *   - missing label => NaN (reference src/iterators/hybrid_reader.c:316-320);
 *   - cosine: stored vectors normalised at add, TopK/Range/Batch normalise a copy of the query,
 *     GetDistanceFrom expects a pre-normalised blob (hybrid_reader.c:295-305);
 *   - range boundary inclusive (reference tests/pytests/test_vecsim.py:2068-2111);
 *   - batches: successive disjoint next-best-n (hybrid_reader.c:387-441).
 * Unpinned choices (SURVEY.md 8c "parity unpinned"), shared with the GPU engine by construction:
 *   - top-K tie-break at rank K: total order (distance, internal row); rows are assigned in
 *     insertion order and a delete moves the LAST row into the hole [upstream-memory D3/D8];
 *   - fp32 summation order: 16 partial sums then a pairwise tree (the shape of an AVX-512 lane
 *     accumulator) -- tolerance-pinned only (1e-6 on 2-d vectors upstream).
 *
 * Pinned by tests/test_oracle_flat.py against the reference's known-answer tests.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { T_F32 = 0, T_F64 = 1, T_BF16 = 2, T_F16 = 3, T_I8 = 4, T_U8 = 5 };
enum { M_L2 = 0, M_IP = 1, M_COS = 2 };

/* ---- half / bfloat16 conversion (round-to-nearest-even on the way down) ---------------------- */
static float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else { /* subnormal: renormalise */
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp + 112) << 23) | (man << 13);
  float f; memcpy(&f, &bits, 4); return f;
}
uint16_t oracle_f32_to_f16(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0));
  if (absx >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* overflows to inf after rounding */
  if (absx < 0x33000001u) return (uint16_t)sign;              /* rounds to zero */
  int e = (int)(absx >> 23) - 127;
  uint32_t man = (absx & 0x7fffffu) | 0x800000u;
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;
  uint32_t half = man >> shift, rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
  if (rem > mid || (rem == mid && (half & 1))) half++;
  if (e < -14) return (uint16_t)(sign | half); /* subnormal (carry into exp is correct) */
  return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (half - 0x400u)));
}
static float bf16_to_f32(uint16_t h) { uint32_t b = (uint32_t)h << 16; float f; memcpy(&f, &b, 4); return f; }
uint16_t oracle_f32_to_bf16(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
  x += 0x7fffu + ((x >> 16) & 1);
  return (uint16_t)(x >> 16);
}
float oracle_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
float oracle_bf16_to_f32(uint16_t h) { return bf16_to_f32(h); }

static size_t type_size(int t) {
  switch (t) { case T_F32: return 4; case T_F64: return 8; case T_BF16: case T_F16: return 2; default: return 1; }
}
/* reference src/iterators/hybrid_reader.c:298-301: INT8/UINT8 cosine blobs carry a float norm */
size_t oracle_blob_size(int type, size_t dim, int metric) {
  size_t s = dim * type_size(type);
  if (metric == M_COS && (type == T_I8 || type == T_U8)) s += sizeof(float);
  return s;
}

/* ---- fp32 reduction in the shape of a 16-lane SIMD accumulator -------------------------------- */
static inline float reduce16(const float *s) {
  float a[8], b[4];
  for (int i = 0; i < 8; i++) a[i] = s[i] + s[i + 8];
  for (int i = 0; i < 4; i++) b[i] = a[i] + a[i + 4];
  return (b[0] + b[2]) + (b[1] + b[3]);
}
#ifdef ORACLE_CLONES
#define HOT __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define HOT
#endif
HOT static float dot_f32(const float *a, const float *b, size_t d) {
  float s[16] = {0};
  size_t i = 0;
  for (; i + 16 <= d; i += 16)
    for (int j = 0; j < 16; j++) s[j] += a[i + j] * b[i + j];
  for (int j = 0; i < d; i++, j++) s[j] += a[i] * b[i];
  return reduce16(s);
}
HOT static float l2_f32(const float *a, const float *b, size_t d) {
  float s[16] = {0};
  size_t i = 0;
  for (; i + 16 <= d; i += 16)
    for (int j = 0; j < 16; j++) { float t = a[i + j] - b[i + j]; s[j] += t * t; }
  for (int j = 0; i < d; i++, j++) { float t = a[i] - b[i]; s[j] += t * t; }
  return reduce16(s);
}

/* fp16 rows: the same 16-lane accumulation over elements widened to fp32.  half -> float is exact, so the F16C
 * hardware conversion (picked at run time where the CPU has it) and the bit-level f16_to_f32 give the same values; lane j
 * still sums the elements i = j (mod 16) in order, products and sums stay separate (no FMA). */
static float dot_or_l2_f16_scalar(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
  float s[16] = {0};
  for (size_t i = 0; i < d; i++) {
    float x = f16_to_f32(a[i]), y = f16_to_f32(b[i]);
    if (l2) { float t = x - y; s[i & 15] += t * t; } else s[i & 15] += x * y;
  }
  return reduce16(s);
}
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,f16c"))) static float dot_or_l2_f16_f16c(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
  __m256 s0 = _mm256_setzero_ps(), s1 = _mm256_setzero_ps();
  size_t i = 0;
  for (; i + 16 <= d; i += 16) {
    __m256 a0 = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(a + i))), a1 = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(a + i + 8)));
    __m256 b0 = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(b + i))), b1 = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(b + i + 8)));
    if (l2) {
      __m256 t0 = _mm256_sub_ps(a0, b0), t1 = _mm256_sub_ps(a1, b1);
      s0 = _mm256_add_ps(s0, _mm256_mul_ps(t0, t0));
      s1 = _mm256_add_ps(s1, _mm256_mul_ps(t1, t1));
    } else {
      s0 = _mm256_add_ps(s0, _mm256_mul_ps(a0, b0));
      s1 = _mm256_add_ps(s1, _mm256_mul_ps(a1, b1));
    }
  }
  float s[16];
  _mm256_storeu_ps(s, s0);
  _mm256_storeu_ps(s + 8, s1);
  for (int j = 0; i < d; i++, j++) {
    float x = f16_to_f32(a[i]), y = f16_to_f32(b[i]);
    if (l2) { float t = x - y; s[j] += t * t; } else s[j] += x * y;
  }
  return reduce16(s);
}
#endif
static float dot_or_l2_f16(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
#if defined(__x86_64__)
  static int have = -1;
  if (have < 0) have = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("f16c");
  if (have) return dot_or_l2_f16_f16c(a, b, d, l2);
#endif
  return dot_or_l2_f16_scalar(a, b, d, l2);
}

/* test hook: 1 when the dispatched fp16 loop and the scalar one agree bit for bit on (a, b) for both metrics */
int oracle_f16_paths_agree(const uint16_t *a, const uint16_t *b, size_t d) {
  for (int l2 = 0; l2 < 2; l2++) {
    float x = dot_or_l2_f16(a, b, d, l2), y = dot_or_l2_f16_scalar(a, b, d, l2);
    if (memcmp(&x, &y, 4) != 0) return 0;
  }
  return 1;
}

/* element i of a blob widened to float (f16/bf16 are up-converted, fp32 accumulate [D5]) */
static inline float elem_f(const void *p, int type, size_t i) {
  switch (type) {
    case T_F32: return ((const float *)p)[i];
    case T_BF16: return bf16_to_f32(((const uint16_t *)p)[i]);
    case T_F16: return f16_to_f32(((const uint16_t *)p)[i]);
    case T_I8: return (float)((const int8_t *)p)[i];
    case T_U8: return (float)((const uint8_t *)p)[i];
    default: return (float)((const double *)p)[i];
  }
}

/* Distance between a stored row `x` and a (pre-normalised for cosine) query `q`. */
double oracle_distance(const void *x, const void *q, size_t d, int type, int metric) {
  if (type == T_F32) {
    if (metric == M_L2) return (double)l2_f32((const float *)x, (const float *)q, d);
    return (double)(1.0f - dot_f32((const float *)x, (const float *)q, d));
  }
  if (type == T_F64) {
    const double *a = x, *b = q; double s = 0;
    if (metric == M_L2) { for (size_t i = 0; i < d; i++) { double t = a[i] - b[i]; s += t * t; } return s; }
    for (size_t i = 0; i < d; i++) s += a[i] * b[i];
    return 1.0 - s;
  }
  if (type == T_I8 || type == T_U8) {
    /* integer accumulate; cosine divides by the two float norms stored behind the elements */
    long long s = 0;
    if (metric == M_L2) {
      for (size_t i = 0; i < d; i++) { int t = (int)elem_f(x, type, i) - (int)elem_f(q, type, i); s += (long long)t * t; }
      return (double)(float)s;
    }
    for (size_t i = 0; i < d; i++) s += (long long)((int)elem_f(x, type, i) * (int)elem_f(q, type, i));
    if (metric == M_IP) return (double)(1.0f - (float)s);
    float na, nb; memcpy(&na, (const char *)x + d, 4); memcpy(&nb, (const char *)q + d, 4);
    return (double)(1.0f - (float)s / (na * nb));
  }
  if (type == T_F16) {
    float r = dot_or_l2_f16((const uint16_t *)x, (const uint16_t *)q, d, metric == M_L2);
    return (double)(metric == M_L2 ? r : 1.0f - r);
  }
  /* bf16 */
  float s[16] = {0};
  for (size_t i = 0; i < d; i++) {
    float a = elem_f(x, type, i), b = elem_f(q, type, i);
    if (metric == M_L2) { float t = a - b; s[i & 15] += t * t; } else s[i & 15] += a * b;
  }
  float r = reduce16(s);
  return (double)(metric == M_L2 ? r : 1.0f - r);
}

/* fp64-accumulated distance: the tolerance yardstick used by the parity tests (never shipped). */
double oracle_distance_f64(const void *x, const void *q, size_t d, int type, int metric) {
  double s = 0;
  for (size_t i = 0; i < d; i++) {
    double a = (type == T_F64) ? ((const double *)x)[i] : (double)elem_f(x, type, i);
    double b = (type == T_F64) ? ((const double *)q)[i] : (double)elem_f(q, type, i);
    if (metric == M_L2) s += (a - b) * (a - b); else s += a * b;
  }
  return metric == M_L2 ? s : 1.0 - s;
}

/* VecSim_Normalize: in-place L2 normalisation (reference src/iterators/hybrid_reader.c:304). */
void oracle_normalize(void *blob, size_t d, int type) {
  if (type == T_F32) {
    float *v = blob; float n = sqrtf(dot_f32(v, v, d));
    for (size_t i = 0; i < d; i++) v[i] /= n;
  } else if (type == T_F64) {
    double *v = blob, n = 0; for (size_t i = 0; i < d; i++) n += v[i] * v[i];
    n = sqrt(n); for (size_t i = 0; i < d; i++) v[i] /= n;
  } else if (type == T_F16 || type == T_BF16) {
    uint16_t *v = blob; float s = 0;
    for (size_t i = 0; i < d; i++) { float a = elem_f(blob, type, i); s += a * a; }
    float n = sqrtf(s);
    for (size_t i = 0; i < d; i++) {
      float a = elem_f(blob, type, i) / n;
      v[i] = (type == T_F16) ? oracle_f32_to_f16(a) : oracle_f32_to_bf16(a);
    }
  } else { /* int8/uint8: elements untouched, norm appended */
    long long s = 0;
    for (size_t i = 0; i < d; i++) { int a = (int)elem_f(blob, type, i); s += (long long)a * a; }
    float n = sqrtf((float)s); memcpy((char *)blob + d, &n, 4);
  }
}

/* ---- the index ---------------------------------------------------------------------------------- */
typedef struct {
  int type, metric, multi;
  size_t dim, stride, block_size;  /* stride = stored bytes per row (incl. int8 cosine norm) */
  size_t n, cap;
  char *rows;      /* row-contiguous; the reference's 1024-row blocks only change allocation */
  size_t *labels;  /* row -> label */
  int last_mode;
} OFlat;

OFlat *oflat_new(int type, size_t dim, int metric, int multi, size_t block_size) {
  if (dim == 0 || type < 0 || type > T_U8 || metric < 0 || metric > M_COS) return NULL;
  OFlat *o = calloc(1, sizeof *o);
  o->type = type; o->metric = metric; o->multi = multi; o->dim = dim;
  o->stride = oracle_blob_size(type, dim, metric);
  o->block_size = block_size ? block_size : 1024;
  return o;
}
void oflat_free(OFlat *o) { if (o) { free(o->rows); free(o->labels); free(o); } }
size_t oflat_size(const OFlat *o) { return o->n; }

static void grow(OFlat *o, size_t need) {
  if (need <= o->cap) return;
  size_t c = o->cap ? o->cap : o->block_size;
  while (c < need) c += (c < (1u << 20)) ? c : (1u << 20);
  o->rows = realloc(o->rows, c * o->stride);
  o->labels = realloc(o->labels, c * sizeof(size_t));
  o->cap = c;
}
static long find_row(const OFlat *o, size_t label, size_t from) {
  for (size_t r = from; r < o->n; r++) if (o->labels[r] == label) return (long)r;
  return -1;
}
static void remove_row(OFlat *o, size_t r) { /* swap-with-last [upstream-memory D8] */
  size_t last = o->n - 1;
  if (r != last) {
    memcpy(o->rows + r * o->stride, o->rows + last * o->stride, o->stride);
    o->labels[r] = o->labels[last];
  }
  o->n--;
}
int oflat_delete(OFlat *o, size_t label) {
  int removed = 0; long r;
  while ((r = find_row(o, label, 0)) >= 0) { remove_row(o, (size_t)r); removed++; if (!o->multi) break; }
  return removed;
}
/* reference src/document.c:721. Single-value: an existing label is overwritten (returns 0). */
int oflat_add(OFlat *o, const void *blob, size_t label) {
  int existed = 0;
  if (!o->multi) existed = oflat_delete(o, label);
  grow(o, o->n + 1);
  char *dst = o->rows + o->n * o->stride;
  memset(dst, 0, o->stride);
  memcpy(dst, blob, o->dim * type_size(o->type));
  if (o->metric == M_COS) oracle_normalize(dst, o->dim, o->type);
  o->labels[o->n++] = label;
  return existed ? 0 : 1;
}
/* bulk append of n raw rows with labels first_label.. (bench / large tests) */
void oflat_add_bulk(OFlat *o, const void *blobs, size_t n, size_t first_label) {
  size_t esz = o->dim * type_size(o->type);
  grow(o, o->n + n);
  for (size_t i = 0; i < n; i++) {
    char *dst = o->rows + o->n * o->stride;
    memset(dst, 0, o->stride);
    memcpy(dst, (const char *)blobs + i * esz, esz);
    if (o->metric == M_COS) oracle_normalize(dst, o->dim, o->type);
    o->labels[o->n++] = first_label + i;
  }
}

/* reference src/iterators/hybrid_reader.c:316: NaN when absent; multi => min over the label's rows */
double oflat_distance_from(const OFlat *o, size_t label, const void *nq) {
  double best = NAN; long r = -1;
  while ((r = find_row(o, label, (size_t)(r + 1))) >= 0) {
    double d = oracle_distance(o->rows + (size_t)r * o->stride, nq, o->dim, o->type, o->metric);
    if (isnan(best) || d < best) best = d;
    if (!o->multi) break;
  }
  return best;
}

static void *prep_query(const OFlat *o, const void *q) {
  void *c = calloc(1, o->stride);
  memcpy(c, q, o->dim * type_size(o->type));
  if (o->metric == M_COS) oracle_normalize(c, o->dim, o->type);
  return c;
}

typedef struct { double d; size_t row; size_t label; } Hit;
static int cmp_dist_row(const void *a, const void *b) {
  const Hit *x = a, *y = b;
  if (x->d < y->d) return -1; if (x->d > y->d) return 1;
  return (x->row > y->row) - (x->row < y->row);
}
static int cmp_dist_label(const void *a, const void *b) {
  const Hit *x = a, *y = b;
  if (x->d < y->d) return -1; if (x->d > y->d) return 1;
  return (x->label > y->label) - (x->label < y->label);
}
static int cmp_label(const void *a, const void *b) {
  const Hit *x = a, *y = b;
  return (x->label > y->label) - (x->label < y->label);
}

/* all rows scored, NaN pushed last, sorted by (distance,row); multi keeps each label's best row */
static Hit *score_all(const OFlat *o, const void *q, size_t *n_out) {
  void *nq = prep_query(o, q);
  Hit *h = malloc((o->n + 1) * sizeof *h);
  for (size_t r = 0; r < o->n; r++) {
    h[r].d = oracle_distance(o->rows + r * o->stride, nq, o->dim, o->type, o->metric);
    if (isnan(h[r].d)) h[r].d = INFINITY; /* NaN sorts last, like the device key order */
    h[r].row = r; h[r].label = o->labels[r];
  }
  free(nq);
  qsort(h, o->n, sizeof *h, cmp_dist_row);
  size_t n = o->n;
  if (o->multi) { /* de-duplicate labels, best (first in order) wins [upstream-memory D10] */
    size_t w = 0;
    for (size_t i = 0; i < n; i++) {
      int dup = 0;
      for (size_t j = 0; j < w; j++) if (h[j].label == h[i].label) { dup = 1; break; }
      if (!dup) h[w++] = h[i];
    }
    n = w;
  }
  *n_out = n;
  return h;
}

/* reference src/iterators/hybrid_reader.c:374. order: 0 BY_SCORE, 1 BY_ID. Returns #hits (<=k). */
size_t oflat_topk(const OFlat *o, const void *q, size_t k, int order, size_t *ids, double *scores) {
  size_t n; Hit *h = score_all(o, q, &n);
  if (k > n) k = n;
  qsort(h, k, sizeof *h, order == 1 ? cmp_label : cmp_dist_label);
  for (size_t i = 0; i < k; i++) { ids[i] = h[i].label; scores[i] = h[i].d; }
  free(h);
  return k;
}

/* reference src/vector_index.c:152; inclusive radius (tests/pytests/test_vecsim.py:2068-2111).
 * ids/scores must hold oflat_size() entries. */
size_t oflat_range(const OFlat *o, const void *q, double radius, int order, size_t *ids, double *scores) {
  size_t n, m = 0; Hit *h = score_all(o, q, &n);
  while (m < n && h[m].d <= radius) m++;
  qsort(h, m, sizeof *h, order == 1 ? cmp_label : cmp_dist_label);
  for (size_t i = 0; i < m; i++) { ids[i] = h[i].label; scores[i] = h[i].d; }
  free(h);
  return m;
}

/* Batch iterator [upstream-memory D7]: all scores on first Next; each Next returns the next-best n. */
typedef struct { Hit *h; size_t n, pos; } OBatch;
OBatch *obatch_new(const OFlat *o, const void *q) {
  OBatch *b = calloc(1, sizeof *b);
  b->h = score_all(o, q, &b->n);
  return b;
}
int obatch_has_next(const OBatch *b) { return b->pos < b->n; }
size_t obatch_next(OBatch *b, size_t n_res, int order, size_t *ids, double *scores) {
  size_t m = b->n - b->pos; if (m > n_res) m = n_res;
  Hit *t = malloc((m + 1) * sizeof *t);
  memcpy(t, b->h + b->pos, m * sizeof *t);
  qsort(t, m, sizeof *t, order == 1 ? cmp_label : cmp_dist_label);
  for (size_t i = 0; i < m; i++) { ids[i] = t[i].label; scores[i] = t[i].d; }
  free(t); b->pos += m;
  return m;
}
void obatch_free(OBatch *b) { if (b) { free(b->h); free(b); } }

/* VecSimIndex_PreferAdHocSearch for BF [upstream-memory D6]; the four decision points the
 * reference pins (tests/pytests/test_vecsim.py:1436-1478,1615-1643,966-968) are checked in
 * tests/test_oracle_flat.py.  Returns 1 for ad-hoc; *mode receives the VecSearchMode recorded. */
/* [upstream-memory D6] BruteForceIndex::preferAdHocSearch: the ratio is subset / LABEL count in float, compared
 * (promoted to double) with double literals; the size cuts use the vector count. */
int oracle_prefer_adhoc2(size_t index_size, size_t label_count, size_t dim, size_t subset, size_t k, int initial_check,
                         int *mode) {
  (void)k;
  if (subset > index_size) subset = index_size; /* an estimate may exceed the index */
  int res;
  float r = index_size ? (float)subset / (float)label_count : 0.0f;
  size_t N = index_size, d = dim;
  if (N <= 5500) res = 1;
  else if (d <= 300) {
    if (r <= 0.15) res = 1;
    else if (r <= 0.35) { if (d <= 75) res = 0; else res = (N <= 550000); }
    else res = 0;
  } else {
    if (r <= 0.55) res = 1;
    else if (d <= 750) res = 0;
    else res = (r <= 0.75);
  }
  if (mode) *mode = res ? (initial_check ? 2 /*HYBRID_ADHOC_BF*/ : 4 /*BATCHES_TO_ADHOC_BF*/) : 3 /*HYBRID_BATCHES*/;
  return res;
}
int oracle_prefer_adhoc(size_t index_size, size_t dim, size_t subset, size_t k, int initial_check, int *mode) {
  return oracle_prefer_adhoc2(index_size, index_size, dim, subset, k, initial_check, mode);
}

/* ---- keyed synthetic corpus: the CPU twin of redisearch_amd/csrc/corpus_kernels.hip ------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): ten rounds of
 * two 32x32->64 multiplies (0xD2511F53, 0xCD9E8D57) with the key bumped by the Weyl constants 0x9E3779B9 / 0xBB67AE85
 * after each round.  Pinned on the Random123 known answers in tests/test_oracle_flat.py. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* rows first_index .. first_index+n-1 of the corpus keyed by `seed`, tightly packed (dim elements of `type` each):
 * element j of row i = word j%4 of Philox(key = seed, counter = (i_lo, i_hi, j/4, 0)); floating types map the word's
 * top 24 bits to [-1,1) on a 2^-23 grid (exact in fp32; f16/bf16 round that to nearest even; f64 widens it), the
 * integer types take the top byte. */
void oracle_philox_rows(uint64_t seed, uint64_t first_index, size_t n, size_t dim, int type, void *out) {
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const size_t esz = type_size(type);
  for (size_t i = 0; i < n; i++) {
    const uint64_t gi = first_index + i;
    uint8_t *row = (uint8_t *)out + i * dim * esz;
    for (size_t q = 0; q * 4 < dim; q++) {
      const uint32_t ctr[4] = {(uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)q, 0u};
      uint32_t w[4];
      oracle_philox4x32_10(ctr, key, w);
      for (size_t e = 0; e < 4 && q * 4 + e < dim; e++) {
        const size_t j = q * 4 + e;
        const float v = (float)(w[e] >> 8) * 0x1p-23f - 1.0f;
        switch (type) {
          case T_F32: ((float *)row)[j] = v; break;
          case T_F64: ((double *)row)[j] = (double)v; break;
          case T_F16: ((uint16_t *)row)[j] = oracle_f32_to_f16(v); break;
          case T_BF16: ((uint16_t *)row)[j] = oracle_f32_to_bf16(v); break;
          default: row[j] = (uint8_t)(w[e] >> 24); break;
        }
      }
    }
  }
}

/* direct pointer to the stored (normalised) rows, for tests that inspect the layout */
const void *oflat_rows(const OFlat *o) { return o->rows; }
const size_t *oflat_labels(const OFlat *o) { return o->labels; }
size_t oflat_stride(const OFlat *o) { return o->stride; }

/* ---- timed scan used ONLY by bench.py's cpu_baseline leg ----------------------------------------
 * One FLAT query the way a single RediSearch worker runs it: every row scored in storage order, a
 * K-bounded max-heap of (distance,label), strict `<` admission [upstream-memory D3]. */
static void heap_sift_down(Hit *h, size_t n, size_t i) {
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && cmp_dist_label(&h[l], &h[m]) > 0) m = l;
    if (r < n && cmp_dist_label(&h[r], &h[m]) > 0) m = r;
    if (m == i) return;
    Hit t = h[i]; h[i] = h[m]; h[m] = t; i = m;
  }
}
size_t oflat_topk_heap(const OFlat *o, const void *q, size_t k, size_t *ids, double *scores) {
  void *nq = prep_query(o, q);
  Hit *h = malloc((k + 1) * sizeof *h); size_t cnt = 0;
  for (size_t r = 0; r < o->n; r++) {
    double d = oracle_distance(o->rows + r * o->stride, nq, o->dim, o->type, o->metric);
    if (cnt < k) {
      h[cnt].d = d; h[cnt].row = r; h[cnt].label = o->labels[r]; cnt++;
      if (cnt == k) for (size_t i = k / 2; i-- > 0;) heap_sift_down(h, k, i);
    } else if (k && d < h[0].d) {
      h[0].d = d; h[0].row = r; h[0].label = o->labels[r]; heap_sift_down(h, k, 0);
    }
  }
  free(nq);
  qsort(h, cnt, sizeof *h, cmp_dist_label);
  for (size_t i = 0; i < cnt; i++) { ids[i] = h[i].label; scores[i] = h[i].d; }
  free(h);
  return cnt;
}
