/*
 * rs_extension.h -- the RediSearch extension (scorer plugin) API, as far as a scoring extension touches it.
 *
 * Boundary 2 of SURVEY.md 8(b): RediSearch loads an extension with dlopen + dlsym("RS_ExtensionInit")
 * (reference src/extension.c:121-145) and hands it an RSExtensionCtx whose RegisterScoringFunction records an
 * RSScoringFunction under an alias (src/extension.c:67-86, API declared at src/redisearch.h:226-287).  The
 * function is then called once per result by rpscoreNext (src/result_processor.c:570-603) with the result tree,
 * the document metadata and the index statistics.
 *
 * A reference extension includes <redisearch.h>; that header drags in redismodule.h and a dozen cheadergen
 * outputs of the Rust workspace, none of which exist on the GPU box.  This header restates exactly the types a
 * SCORER reads, with the reference's names and -- because the structs are shared memory between module and
 * plugin -- the reference's LAYOUT.  tests/test_scorer_plugin.py compiles a probe against the reference's own
 * headers (where /root/reference exists) and compares every sizeof/offsetof below.
 *
 * The result-tree accessors (IndexResult_*, AggregateResult_*, QueryTerm_*) are implemented in Rust inside the
 * module (src/redisearch_rs/headers/types_ffi.h, query_term_ffi.h); a plugin leaves them undefined and the
 * dynamic linker binds them to the module's at load time.
 */
#ifndef RSGPU_RS_EXTENSION_H
#define RSGPU_RS_EXTENSION_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REDISEARCH_OK 0  /* reference src/redisearch.h:22 */
#define REDISEARCH_ERR 1 /* reference src/redisearch.h:23 */

typedef uint64_t t_docId;        /* reference src/redisearch_rs/headers/rqe_core.h */
typedef __uint128_t t_fieldMask; /* 64-bit targets: rqe_core.h:19 */

/* ---- document metadata (reference src/redisearch.h:97-132) ------------------------------------------------ */
typedef struct {
  char *data;
  size_t len;
} RSPayload; /* src/redisearch.h:69-72 */

#define RS_DOCUMENT_HAS_PAYLOAD 0x02 /* Document_HasPayload, src/redisearch.h:77 */

typedef struct RSDocumentMetadata_s {
  t_docId id;
  char *keyPtr;
  float score;               /* a-priori document score: every scorer multiplies by it */
  uint32_t maxTermFreq : 24; /* TFIDF normaliser */
  uint32_t flags : 8;
  uint32_t docLen : 24;      /* TFIDF.DOCNORM normaliser, BM25STD length term */
  uint32_t type : 8;
  uint16_t ref_count;
  int64_t expirationTimeNs;
  void *sortVector;          /* RSSortingVector: one pointer (sorting_vector.h) */
  struct RSByteOffsets *byteOffsets;
  struct RSDocumentMetadata_s *nextInChain;
  RSPayload *payload;
} RSDocumentMetadata;

/* ---- result tree (reference src/redisearch_rs/headers/index_result_rs.h:520-640) --------------------------- */
enum {
  RSResultData_Union = 1,
  RSResultData_Intersection = 2,
  RSResultData_Term = 4,
  RSResultData_Virtual = 8,
  RSResultData_Numeric = 16,
  RSResultData_Metric = 32,
  RSResultData_HybridMetric = 64,
};

typedef struct RSQueryTerm RSQueryTerm; /* opaque: query_term.h:19 */

/* RawAggregateResult_Active: a tag byte, a thin-vec pointer owned by Rust, the kind mask (24 bytes). Opaque to C:
 * read through AggregateResult_*. */
typedef struct {
  uint8_t tag;
  void *records;
  uint8_t kind_mask;
} RSAggregateResult;

/* RawTermRecord_Active (32 bytes): tag, the query term, the encoded offsets (borrowed). */
typedef struct {
  uint8_t tag;
  const RSQueryTerm *term;
  struct {
    const uint8_t *data;
    uint32_t len;
  } offsets;
} RSTermRecord;

typedef struct {
  uint8_t tag; /* one of RSResultData_* */
  union {
    RSAggregateResult agg; /* Union / Intersection / HybridMetric */
    RSTermRecord term;     /* Term */
    double num;            /* Numeric / Metric */
  };
} RSResultData;

typedef struct RSIndexResult {
  t_docId docId;
  const RSDocumentMetadata *dmd;
  t_fieldMask fieldMask;
  uint32_t freq; /* total frequency of the records below */
  RSResultData data;
  void *metrics; /* MetricsVec: one pointer */
  double weight; /* the producing iterator's weight */
  bool hasFieldExpiration;
} RSIndexResult;

typedef struct AggregateRecordsSlice {
  const RSIndexResult *const *ptr;
  size_t len;
} AggregateRecordsSlice; /* types_ffi.h:35-38 */

/* Implemented by the module (Rust); undefined in a plugin.  reference types_ffi.h:120,141,179,222,295;
 * query_term_ffi.h:52,74,96. */
const RSIndexResult *AggregateResult_Get(const RSAggregateResult *agg, size_t index);
AggregateRecordsSlice AggregateResult_GetRecordsSlice(const RSAggregateResult *agg);
size_t AggregateResult_NumChildren(const RSAggregateResult *agg);
const RSAggregateResult *IndexResult_AggregateRefUnchecked(const RSIndexResult *result);
RSQueryTerm *IndexResult_QueryTermRef(const RSIndexResult *result);
double QueryTerm_GetBM25_IDF(const RSQueryTerm *term);
double QueryTerm_GetIDF(const RSQueryTerm *term);
const char *QueryTerm_GetStrAndLen(const RSQueryTerm *term, size_t *out_len);

/* ---- EXPLAINSCORE tree (reference src/score_explain.h:20-24) ---------------------------------------------- */
typedef struct RSScoreExplain {
  char *str;
  int numChildren;
  struct RSScoreExplain *children; /* one array of numChildren nodes */
} RSScoreExplain;

/* ---- scoring function API (reference src/redisearch.h:226-287) -------------------------------------------- */
#define RS_SCORE_FILTEROUT (-1.0 / 0.0)

typedef struct {
  size_t numDocs;
  size_t numTerms;
  double avgDocLen;
} RSIndexStats;

typedef struct {
  void *extdata;      /* privdata given at registration */
  const void *qdata;  /* query payload */
  size_t qdatalen;
  RSIndexStats indexStats;
  void *scrExp;       /* RSScoreExplain* when EXPLAINSCORE was asked, else NULL */
  int (*GetSlop)(const RSIndexResult *res); /* = IndexResult_MinOffsetDelta (src/extension.c:161) */
  uint64_t tanhFactor;
} ScoringFunctionArgs;

typedef double (*RSScoringFunction)(const ScoringFunctionArgs *ctx, const RSIndexResult *res,
                                    const RSDocumentMetadata *dmd, double minScore);
typedef void (*RSFreeFunction)(void *);

/* Query expanders are not part of this path; the member keeps the struct's layout. */
typedef int (*RSQueryTokenExpander)(void *ctx, void *token);

typedef struct RSExtensionCtx {
  int (*RegisterScoringFunction)(const char *alias, RSScoringFunction func, RSFreeFunction ff, void *privdata);
  int (*RegisterQueryExpander)(const char *alias, RSQueryTokenExpander exp, RSFreeFunction ff, void *privdata);
} RSExtensionCtx;

typedef int (*RSExtensionInitFunc)(RSExtensionCtx *ctx);

/* Scorer aliases (reference src/redisearch_rs/headers/query_types.h:504-573). */
#define TFIDF_SCORER_NAME "TFIDF"
#define TFIDF_DOCNORM_SCORER_NAME "TFIDF.DOCNORM"
#define BM25_SCORER_NAME "BM25"
#define BM25_STD_SCORER_NAME "BM25STD"
#define BM25_STD_NORMALIZED_TANH_SCORER_NAME "BM25STD.TANH"
#define BM25_STD_NORMALIZED_MAX_SCORER_NAME "BM25STD.NORM"
#define DISMAX_SCORER_NAME "DISMAX"
#define DOCSCORE_SCORER "DOCSCORE"
#define HAMMINGDISTANCE_SCORER "HAMMING"

/* ---- what librsgpu_scorers.so exports (redisearch_amd/csrc/scorer_plugin.c) ---------------------------------
 * RS_ExtensionInit registers the nine default scorers.  Inside a stock module the default aliases are taken by
 * DefaultExtensionInit (src/ext/default.c:739-784) and a duplicate is refused (src/extension.c:79-82), so every
 * alias that is refused is registered as "RSGPU.<alias>" instead (`SCORER RSGPU.BM25STD`); a module built
 * without the default scorers gets the plain names.  REDISEARCH_ERR only if an alias is refused under both names. */
int RS_ExtensionInit(RSExtensionCtx *ctx);
#define RSGPU_SCORER_ALIAS_PREFIX "RSGPU."

/* EXPLAINSCORE strings are released by the module with its own allocator (SEDestroy, src/score_explain.c:35-52).
 * By default the plugin looks up RedisModule_Calloc / RedisModule_Free in the process at first use and falls
 * back to calloc / free; a host that knows better installs its pair here before the first query. */
void RSGPU_Scorers_SetAllocator(void *(*calloc_fn)(size_t, size_t), void (*free_fn)(void *));

#ifdef __cplusplus
}
#endif
#endif /* RSGPU_RS_EXTENSION_H */
