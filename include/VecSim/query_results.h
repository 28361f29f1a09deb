/*
 * VecSim/query_results.h -- reply / result-iterator half of the VecSim C ABI.
 *
 * Replaces deps/VectorSimilarity/src/VecSim/query_results.h (absent submodule).  The symbol set is
 * exactly the bindgen allowlist at reference src/redisearch_rs/ffi/build.rs:33-51 plus the two
 * extra calls made from C (VecSimQueryReply_Len, VecSimQueryReply_IteratorHasNext).
 *
 * Ownership (reference src/redisearch_rs/c_wrappers/vecsim/src/reply.rs:120-131,176-181):
 *   - a VecSimQueryReply is owned by the caller and released with VecSimQueryReply_Free;
 *   - an iterator borrows the reply; free the iterator first;
 *   - a VecSimQueryResult* returned by IteratorNext is borrowed from the reply;
 *   - both Free functions accept NULL (reference src/iterators/hybrid_reader.c:543-544,579-580
 *     call them on never-assigned fields).
 */
#ifndef VECSIM_QUERY_RESULTS_H
#define VECSIM_QUERY_RESULTS_H

#include "vec_sim_common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Requested ordering of a reply. BY_SCORE_THEN_ID is internal-only upstream
 * (reference src/redisearch_rs/c_wrappers/vecsim/src/params.rs:12-35). */
typedef enum { BY_SCORE, BY_ID, BY_SCORE_THEN_ID } VecSimQueryReply_Order;

typedef struct VecSimQueryResult VecSimQueryResult;
typedef struct VecSimQueryReply VecSimQueryReply;
typedef struct VecSimQueryReply_Iterator VecSimQueryReply_Iterator;
typedef struct VecSimBatchIterator VecSimBatchIterator;

/* label (== doc id) of one hit; reference src/iterators/hybrid_reader.c:63,83 */
size_t VecSimQueryResult_GetId(const VecSimQueryResult *item);
/* distance of one hit, widened to double; reference src/iterators/hybrid_reader.c:70,84 */
double VecSimQueryResult_GetScore(const VecSimQueryResult *item);

/* number of hits; reference src/vector_index.c:101 */
size_t VecSimQueryReply_Len(VecSimQueryReply *reply);
/* OK or TimedOut; reference src/iterators/hybrid_reader.c:376,418, src/vector_index.c:153 */
VecSimQueryReply_Code VecSimQueryReply_GetCode(VecSimQueryReply *reply);
/* NULL-safe */
void VecSimQueryReply_Free(VecSimQueryReply *reply);

/* reference src/iterators/hybrid_reader.c:375,422 */
VecSimQueryReply_Iterator *VecSimQueryReply_GetIterator(VecSimQueryReply *reply);
/* returns NULL once exhausted (reference src/redisearch_rs/c_wrappers/vecsim/src/reply.rs:163-166) */
VecSimQueryResult *VecSimQueryReply_IteratorNext(VecSimQueryReply_Iterator *iterator);
bool VecSimQueryReply_IteratorHasNext(VecSimQueryReply_Iterator *iterator);
void VecSimQueryReply_IteratorReset(VecSimQueryReply_Iterator *iterator);
/* NULL-safe */
void VecSimQueryReply_IteratorFree(VecSimQueryReply_Iterator *iterator);

/* Next disjoint batch of the `n_results` next-best hits, ordered as requested (the hybrid
 * iterator asks BY_ID for its merge-join, reference src/iterators/hybrid_reader.c:417). */
VecSimQueryReply *VecSimBatchIterator_Next(VecSimBatchIterator *iterator, size_t n_results,
                                           VecSimQueryReply_Order order);
/* reference src/iterators/hybrid_reader.c:398 */
bool VecSimBatchIterator_HasNext(VecSimBatchIterator *iterator);
/* reference src/iterators/hybrid_reader.c:433,441 */
void VecSimBatchIterator_Free(VecSimBatchIterator *iterator);
void VecSimBatchIterator_Reset(VecSimBatchIterator *iterator);

#ifdef __cplusplus
}
#endif
#endif /* VECSIM_QUERY_RESULTS_H */
