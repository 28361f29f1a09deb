/*
 * VecSim/vec_sim.h -- index half of the VecSim C ABI, served by the MI355X FLAT engine.
 *
 * Replaces deps/VectorSimilarity/src/VecSim/vec_sim.h (absent submodule).  Every entry point cites
 * the reference call site that binds it.  Only VecSimAlgo_BF ("FLAT") indexes are served; the
 * corpus lives row-contiguous in HBM and TopK/Range/Batch/GetDistanceFrom run as HIP kernels
 * (DESIGN.md).  No CPU fallback exists: a process without a usable gfx950 device gets NULL from
 * VecSimIndex_New and a logged error.
 *
 * Threading contract honoured: global setters once at init; many concurrent query
 * threads; a single writer (AddVector/DeleteVector) excluded from readers by the caller's lock.
 */
#ifndef VECSIM_VEC_SIM_H
#define VECSIM_VEC_SIM_H

#include "vec_sim_common.h"
#include "query_results.h"
#include "info_iterator.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VecSimIndex VecSimIndex;
typedef struct VecSimAdhocBfCtx VecSimAdhocBfCtx;

/* ---- lifecycle -------------------------------------------------------------------------------- */
/* reference src/vector_index.c:89. NULL on failure (non-BF algo, dim==0, unsupported type, no GPU). */
VecSimIndex *VecSimIndex_New(const VecSimParams *params);
/* Disk (Flex) variant: never served by this engine; returns NULL. */
VecSimIndex *VecSimIndex_NewDisk(const VecSimParamsDisk *params);
/* reference src/field_spec.c:49 */
void VecSimIndex_Free(VecSimIndex *index);

/* reference src/spec.c:623 / :613 */
size_t VecSimIndex_EstimateInitialSize(const VecSimParams *params);
size_t VecSimIndex_EstimateElementSize(const VecSimParams *params);

/* ---- writes ----------------------------------------------------------------------------------- */
/* reference src/document.c:721. blob = dim*sizeof(type) bytes, borrowed for the call.
 * Returns the number of NEW vectors (1, or 0 when an existing single-value label is overwritten). */
int VecSimIndex_AddVector(VecSimIndex *index, const void *blob, size_t label);
/* reference src/indexer.c:186, src/spec.c:3539. Returns the number of vectors removed. */
int VecSimIndex_DeleteVector(VecSimIndex *index, size_t label);

/* ---- info ------------------------------------------------------------------------------------- */
/* reference src/iterators/hybrid_reader.c:359,392,400 */
size_t VecSimIndex_IndexSize(VecSimIndex *index);
/* reference src/vector_index.c:241 */
VecSimIndexBasicInfo VecSimIndex_BasicInfo(VecSimIndex *index);
/* reference src/info/field_spec_info.c:281 */
VecSimIndexStatsInfo VecSimIndex_StatsInfo(VecSimIndex *index);
/* reference src/debug_commands.c:1714 */
VecSimDebugInfoIterator *VecSimIndex_DebugInfoIterator(VecSimIndex *index);
/* the iterator's own accessors: VecSim/info_iterator.h */

/* ---- queries (the hot path) -------------------------------------------------------------------- */
/* reference src/vector_index.c:744. Fills qparams from raw (name,value) pairs. */
VecSimResolveCode VecSimIndex_ResolveParams(VecSimIndex *index, VecSimRawParam *rparams, int paramNum,
                                            VecSimQueryParams *qparams, VecsimQueryType query_type);

/* reference src/iterators/hybrid_reader.c:374 -- exact K-NN over every stored vector.
 * queryBlob is dim*sizeof(type) bytes, borrowed; a cosine query is normalised internally. */
VecSimQueryReply *VecSimIndex_TopKQuery(VecSimIndex *index, const void *queryBlob, size_t k,
                                        VecSimQueryParams *queryParams, VecSimQueryReply_Order order);

/* reference src/vector_index.c:152 -- all vectors with distance <= radius (inclusive). */
VecSimQueryReply *VecSimIndex_RangeQuery(VecSimIndex *index, const void *queryBlob, double radius,
                                         VecSimQueryParams *queryParams, VecSimQueryReply_Order order);

/* reference src/iterators/hybrid_reader.c:316 -- distance from the stored vector `label` to a
 * query blob that the CALLER has already normalised for cosine (hybrid_reader.c:295-305).
 * NaN when the label is absent. */
double VecSimIndex_GetDistanceFrom_Unsafe(VecSimIndex *index, size_t label, const void *blob);

/* reference src/iterators/hybrid_reader.c:369,684 -- ad-hoc BF vs batches heuristic. */
bool VecSimIndex_PreferAdHocSearch(VecSimIndex *index, size_t subsetSize, size_t k, bool initialCheck);

/* reference src/iterators/hybrid_reader.c:387. The query blob is copied. */
VecSimBatchIterator *VecSimBatchIterator_New(VecSimIndex *index, const void *queryBlob,
                                             VecSimQueryParams *queryParams);

/* reference src/iterators/hybrid_reader.c:214-266 (disk path upstream; here the batched GPU gather
 * for any FLAT index).  _New normalises a copy of the query for cosine. */
VecSimAdhocBfCtx *VecSimIndex_AdhocBfCtx_New(VecSimIndex *index, const void *queryBlob);
double VecSimIndex_AdhocBfCtx_GetDistanceFrom(VecSimAdhocBfCtx *ctx, size_t label);
void VecSimIndex_AdhocBfCtx_GetExactDistances(VecSimAdhocBfCtx *ctx, const size_t *labels,
                                              double *distances_out, size_t count);
void VecSimIndex_AdhocBfCtx_Free(VecSimAdhocBfCtx *ctx);

/* reference src/iterators/hybrid_reader.c:307,328 and src/vector_index.c:832 -- no-ops on FLAT. */
void VecSimTieredIndex_AcquireSharedLocks(VecSimIndex *index);
void VecSimTieredIndex_ReleaseSharedLocks(VecSimIndex *index);
void VecSimTieredIndex_GC(VecSimIndex *index);

/* HNSW-only debug helpers (reference src/debug_commands.c:1756-1779): VecSim/vec_sim_debug.h. */

/* ---- blob helpers ------------------------------------------------------------------------------ */
/* reference src/iterators/hybrid_reader.c:304 -- in-place L2 normalisation. */
void VecSim_Normalize(void *blob, size_t dim, VecSimType type);
/* reference src/iterators/hybrid_reader.c:301 -- dim*sizeof(type) (+sizeof(float) norm slot for
 * INT8/UINT8 cosine). */
size_t VecSimParams_GetQueryBlobSize(VecSimType type, size_t dim, VecSimMetric metric);

/* ---- process-wide hooks (reference src/module-init/module-init.c:147-151) ----------------------- */
void VecSim_SetMemoryFunctions(VecSimMemoryFunctions memoryfunctions);
void VecSim_SetTimeoutCallbackFunction(timeoutCallbackFunction callback);
void VecSim_SetLogCallbackFunction(logCallbackFunction callback);
void VecSim_SetWriteMode(VecSimWriteMode mode);
/* reference src/util/workers.c:58,104 */
void VecSim_UpdateThreadPoolSize(size_t new_size);
/* reference src/info/info_command.c:299 */
size_t VecSim_GetSharedMemory(void);

#ifdef __cplusplus
}
#endif
#endif /* VECSIM_VEC_SIM_H */
