/* Plain-C client of the drop-in library: the calls hybrid_reader.c / vector_index.c make, in their order
 * (reference src/vector_index.c:89 VecSimIndex_New, src/document.c:721 VecSimIndex_AddVector,
 *  src/iterators/hybrid_reader.c:374 VecSimIndex_TopKQuery, :61-85 reply iteration, :543-544 frees).
 *
 *   gcc -Iinclude examples/knn_example.c -Lredisearch_amd/lib -lVectorSimilarity -Wl,-rpath,$PWD/redisearch_amd/lib -o knn_example
 *
 * Prints the 5 nearest of 1000 vectors [i,i,i,i] to the query [100,100,100,100] (L2): ids 100, 99, 101, 98, 102. */
#include <stdio.h>
#include <string.h>

#include "VecSim/query_results.h"
#include "VecSim/vec_sim.h"
#include "rsgpu_ext.h"

int main(void) {
  VecSimParams params;
  memset(&params, 0, sizeof params);
  params.algo = VecSimAlgo_BF;
  params.algoParams.bfParams.type = VecSimType_FLOAT32;
  params.algoParams.bfParams.dim = 4;
  params.algoParams.bfParams.metric = VecSimMetric_L2;
  params.algoParams.bfParams.blockSize = 1024;
  VecSimIndex *index = VecSimIndex_New(&params);
  if (!index) {
    fprintf(stderr, "VecSimIndex_New failed: %s\n", RSGPU_LastError());
    return 2;
  }
  for (size_t i = 1; i <= 1000; i++) {
    float v[4] = {(float)i, (float)i, (float)i, (float)i};
    VecSimIndex_AddVector(index, v, i);
  }
  float q[4] = {100.f, 100.f, 100.f, 100.f};
  VecSimQueryReply *reply = VecSimIndex_TopKQuery(index, q, 5, NULL, BY_SCORE);
  if (!reply || VecSimQueryReply_GetCode(reply) != VecSim_QueryReply_OK) return 3;
  VecSimQueryReply_Iterator *it = VecSimQueryReply_GetIterator(reply);
  int rc = 0;
  static const size_t want[5] = {100, 99, 101, 98, 102};
  for (int i = 0; VecSimQueryReply_IteratorHasNext(it); i++) {
    VecSimQueryResult *r = VecSimQueryReply_IteratorNext(it);
    printf("%zu %.1f\n", VecSimQueryResult_GetId(r), VecSimQueryResult_GetScore(r));
    if (i >= 5 || VecSimQueryResult_GetId(r) != want[i]) rc = 4;
  }
  VecSimQueryReply_IteratorFree(it);
  VecSimQueryReply_Free(reply);
  VecSimIndex_Free(index);
  return rc;
}
