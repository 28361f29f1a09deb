/* Prints the layout facts a scorer extension depends on.  Compiled twice by tests/test_scorer_plugin.py: against the
 * reference's own headers (-DPROBE_REFERENCE, where /root/reference exists) and against include/rs_extension.h; the two
 * outputs must be identical. */
#include <stdio.h>
#include <stddef.h>
#include <string.h>
#ifdef PROBE_REFERENCE
#include "redisearch.h"
#include "types_ffi.h"
#include "index_result_rs.h"
#include "score_explain.h"
#include "query_types.h"
#define AGG_MEMBER union_
#else
#include "rs_extension.h"
#define AGG_MEMBER agg
#endif
#define P(x) printf(#x " %zu\n", (size_t)(x))
int main(void) {
  P(sizeof(RSIndexResult));
  P(offsetof(RSIndexResult, docId));
  P(offsetof(RSIndexResult, dmd));
  P(offsetof(RSIndexResult, fieldMask));
  P(offsetof(RSIndexResult, freq));
  P(offsetof(RSIndexResult, data));
  P(offsetof(RSIndexResult, metrics));
  P(offsetof(RSIndexResult, weight));
  P(offsetof(RSIndexResult, hasFieldExpiration));
  P(sizeof(RSResultData));
  P(sizeof(RSAggregateResult));
  P(sizeof(RSTermRecord));
  RSIndexResult r;
  memset(&r, 0, sizeof r);
  printf("data.tag %zu\n", (size_t)((char *)&r.data.tag - (char *)&r.data));
  printf("data.aggregate %zu\n", (size_t)((char *)&r.data.AGG_MEMBER - (char *)&r.data));
  printf("data.term %zu\n", (size_t)((char *)&r.data.term - (char *)&r.data));
  printf("tags %d %d %d %d %d %d %d\n", RSResultData_Union, RSResultData_Intersection, RSResultData_Term,
         RSResultData_Virtual, RSResultData_Numeric, RSResultData_Metric, RSResultData_HybridMetric);
  P(sizeof(RSDocumentMetadata));
  P(offsetof(RSDocumentMetadata, id));
  P(offsetof(RSDocumentMetadata, keyPtr));
  P(offsetof(RSDocumentMetadata, score));
  P(offsetof(RSDocumentMetadata, ref_count));
  P(offsetof(RSDocumentMetadata, expirationTimeNs));
  P(offsetof(RSDocumentMetadata, sortVector));
  P(offsetof(RSDocumentMetadata, byteOffsets));
  P(offsetof(RSDocumentMetadata, nextInChain));
  P(offsetof(RSDocumentMetadata, payload));
  RSDocumentMetadata d;
  memset(&d, 0, sizeof d);
  d.maxTermFreq = 0xABCDEF;
  d.flags = 0x5A;
  d.docLen = 0x123456;
  d.type = 0x3;
  printf("bitfields");
  for (size_t i = offsetof(RSDocumentMetadata, score) + 4; i < offsetof(RSDocumentMetadata, ref_count); i++)
    printf(" %02x", ((unsigned char *)&d)[i]);
  printf("\n");
  P(sizeof(RSPayload));
  P(offsetof(RSPayload, len));
  P(sizeof(ScoringFunctionArgs));
  P(offsetof(ScoringFunctionArgs, extdata));
  P(offsetof(ScoringFunctionArgs, qdata));
  P(offsetof(ScoringFunctionArgs, qdatalen));
  P(offsetof(ScoringFunctionArgs, indexStats));
  P(offsetof(ScoringFunctionArgs, scrExp));
  P(offsetof(ScoringFunctionArgs, GetSlop));
  P(offsetof(ScoringFunctionArgs, tanhFactor));
  P(sizeof(RSIndexStats));
  P(offsetof(RSIndexStats, numDocs));
  P(offsetof(RSIndexStats, avgDocLen));
  P(sizeof(RSScoreExplain));
  P(offsetof(RSScoreExplain, numChildren));
  P(offsetof(RSScoreExplain, children));
  P(sizeof(RSExtensionCtx));
  P(offsetof(RSExtensionCtx, RegisterQueryExpander));
  P(sizeof(AggregateRecordsSlice));
  P(offsetof(AggregateRecordsSlice, len));
  printf("consts %d %d %s %s %s %s %s %s %s %s %s\n", REDISEARCH_OK, REDISEARCH_ERR, TFIDF_SCORER_NAME,
         TFIDF_DOCNORM_SCORER_NAME, BM25_SCORER_NAME, BM25_STD_SCORER_NAME, BM25_STD_NORMALIZED_TANH_SCORER_NAME,
         BM25_STD_NORMALIZED_MAX_SCORER_NAME, DISMAX_SCORER_NAME, DOCSCORE_SCORER, HAMMINGDISTANCE_SCORER);
  return 0;
}
