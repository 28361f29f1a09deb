/* Plain-C client of the hybrid path: `hello (world|words) -spam` as the filter of a hybrid query -- BM25STD top-5 next to the 4
 * nearest vectors among the hits -- in ONE call (RSGPU_HybridTreeQuery, include/rsgpu_search.h).  What the reference does with
 * an iterator tree handed to its hybrid iterator (src/iterators/hybrid_reader.c:625 NewHybridVectorIterator(childIt = ...),
 * rqe_iterators/src/intersection.rs:94-119, union_flat.rs, not.rs) plus the scoring loop behind it (result_processor.c:570-603).
 *
 *   gcc -Iinclude examples/hybrid_tree_example.c -Lredisearch_amd/lib -lVectorSimilarity -Wl,-rpath,$PWD/redisearch_amd/lib -o hybrid_tree_example
 *
 * 200 000 documents; term lists by divisibility (hello: 2 | id, world: 3 | id, words: 5 | id, spam: 7 | id), uploaded in the raw
 * doc-id codec (u32 deltas from the block's first doc id, blocks of 100); documents 1 .. 20 000 carry the vector [id, id, id, id].
 * Prints: the hit count (checked against a host loop), the path (2 = the general tile kernel), the top-5 and the KNN answer
 * (3000, 3006, 2994, 2990: 3010 is nearer than 2990 but holds `spam`). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "VecSim/vec_sim.h"
#include "rsgpu_ext.h"
#include "rsgpu_search.h"

#define N_DOCS 200000u
#define BLOCK 100u

static RSGPU_Postings *upload_multiples(unsigned step, size_t *n_out) {
  const size_t n = N_DOCS / step, nb = (n + BLOCK - 1) / BLOCK;
  uint64_t *first = malloc(nb * 8), *last = malloc(nb * 8), *off = malloc((nb + 1) * 8);
  uint32_t *cnt = malloc(nb * 4), *bytes = malloc(n * 4);
  for (size_t b = 0; b < nb; b++) {
    const size_t e0 = b * BLOCK, e1 = e0 + BLOCK < n ? e0 + BLOCK : n;
    first[b] = (uint64_t)(e0 + 1) * step;
    last[b] = (uint64_t)e1 * step;
    cnt[b] = (uint32_t)(e1 - e0);
    off[b] = e0 * 4;
    for (size_t e = e0; e < e1; e++) bytes[e] = (uint32_t)((e + 1) * step - first[b]);
  }
  off[nb] = n * 4;
  RSGPU_Postings *p = RSGPU_Postings_Upload(RSGPU_CODEC_RAW_DOCIDS, nb, first, last, cnt, off, (const uint8_t *)bytes);
  free(first); free(last); free(off); free(cnt); free(bytes);
  *n_out = n;
  return p;
}

int main(void) {
  size_t n[4];
  RSGPU_Postings *lists[4];
  static const unsigned step[4] = {2, 3, 5, 7};
  for (int l = 0; l < 4; l++) {
    lists[l] = upload_multiples(step[l], &n[l]);
    if (!lists[l]) {
      fprintf(stderr, "RSGPU_Postings_Upload failed: %s\n", RSGPU_LastError());
      return 2;
    }
  }
  /* the document table: length 50 + id % 100, score 1 */
  uint32_t *doc_len = malloc((N_DOCS + 1) * 4);
  float *doc_score = malloc((N_DOCS + 1) * 4);
  for (unsigned i = 0; i <= N_DOCS; i++) {
    doc_len[i] = 50 + i % 100;
    doc_score[i] = 1.0f;
  }
  RSGPU_DocTable *table = RSGPU_DocTable_Upload(N_DOCS + 1, doc_len, doc_score, NULL);
  /* the FLAT index: documents 1 .. 20 000 have a vector */
  VecSimParams params;
  memset(&params, 0, sizeof params);
  params.algo = VecSimAlgo_BF;
  params.algoParams.bfParams.type = VecSimType_FLOAT32;
  params.algoParams.bfParams.dim = 4;
  params.algoParams.bfParams.metric = VecSimMetric_L2;
  params.algoParams.bfParams.blockSize = 1024;
  VecSimIndex *index = VecSimIndex_New(&params);
  if (!table || !index) {
    fprintf(stderr, "setup failed: %s\n", RSGPU_LastError());
    return 2;
  }
  for (size_t i = 1; i <= 20000; i++) {
    float v[4] = {(float)i, (float)i, (float)i, (float)i};
    VecSimIndex_AddVector(index, v, i);
  }

  /* hello (world|words) -spam */
  size_t first[] = {0, 1, 3, 4};
  int op[] = {RSGPU_OP_TERM, RSGPU_OP_UNION, RSGPU_OP_NOT};
  RSGPU_TreeQuery tree = {RSGPU_OP_INTERSECT, 3, first, op, /*group_weight*/ NULL, lists, /*max_slop*/ -1, /*in_order*/ 0};
  double idf[4] = {0, 0, 0, 0}, bidf[4], weight[4] = {1.0, 1.0, 1.0, 0.0};
  for (int l = 0; l < 4; l++) bidf[l] = l < 3 ? RSGPU_CalculateIDF_BM25(N_DOCS, n[l]) : 0.0;
  RSGPU_ScoreArgs sa;
  memset(&sa, 0, sizeof sa);
  sa.scorer = RSGPU_SCORER_BM25STD;
  sa.num_docs = N_DOCS;
  sa.avg_doc_len = 99.5;
  sa.root_weight = 1.0;
  sa.idf = idf;
  sa.bm25_idf = bidf;
  sa.weight = weight;
  float q[4] = {3000.2f, 3000.2f, 3000.2f, 3000.2f};
  uint64_t top_ids[5], knn_ids[4];
  double top_scores[5], knn_dists[4];
  RSGPU_HybridQueryArgs a;
  memset(&a, 0, sizeof a);
  a.table = table;
  a.score = &sa;
  a.top_n = 5;
  a.index = index;
  a.query = q;
  a.k = 4;
  a.top_ids = top_ids;
  a.top_scores = top_scores;
  a.knn_ids = knn_ids;
  a.knn_dists = knn_dists;
  if (RSGPU_HybridTreeQuery(&tree, &a) != 0) {
    fprintf(stderr, "RSGPU_HybridTreeQuery failed: %s\n", RSGPU_LastError());
    return 3;
  }
  int rc = 0;
  size_t want_hits = 0;
  for (unsigned i = 1; i <= N_DOCS; i++) want_hits += (i % 2 == 0 && (i % 3 == 0 || i % 5 == 0) && i % 7 != 0) ? 1 : 0;
  printf("hits %zu path %d\n", a.n_hits, RSGPU_HybridQueryPath());
  if (a.n_hits != want_hits || a.n_top != 5 || a.n_knn != 4) rc = 4;
  for (size_t i = 0; i < a.n_top; i++) {
    const uint64_t d = top_ids[i];
    printf("top %llu %.17g\n", (unsigned long long)d, top_scores[i]);
    if (!(d % 2 == 0 && (d % 3 == 0 || d % 5 == 0) && d % 7 != 0)) rc = 5;            /* every winner is a hit */
    if (i && !(top_scores[i] < top_scores[i - 1] || (top_scores[i] == top_scores[i - 1] && d > top_ids[i - 1]))) rc = 6;
  }
  static const uint64_t want_knn[4] = {3000, 3006, 2994, 2990};
  for (size_t i = 0; i < a.n_knn; i++) {
    printf("knn %llu %.4f\n", (unsigned long long)knn_ids[i], knn_dists[i]);
    if (i >= 4 || knn_ids[i] != want_knn[i]) rc = 7;
  }
  VecSimIndex_Free(index);
  RSGPU_DocTable_Free(table);
  for (int l = 0; l < 4; l++) RSGPU_Postings_Free(lists[l]);
  free(doc_len);
  free(doc_score);
  return rc;
}
