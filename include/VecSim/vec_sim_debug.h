/*
 * VecSim/vec_sim_debug.h -- debug-only entry points of the VecSim C ABI.
 *
 * Replaces deps/VectorSimilarity/src/VecSim/vec_sim_debug.h (absent submodule); included by
 * reference src/debug_commands.c:21.  Both functions serve FT.DEBUG DUMP_HNSW (replyDumpHNSW,
 * reference src/debug_commands.c:1756-1779), which the caller only reaches after checking
 * VecSimIndex_BasicInfo(index).algo == VecSimAlgo_HNSWLIB (:1814-1819).  This engine serves FLAT
 * indexes only, so Get... always answers VecSimDebugCommandCode_BadIndex and leaves *neighborsData NULL.
 */
#ifndef VECSIM_VEC_SIM_DEBUG_H
#define VECSIM_VEC_SIM_DEBUG_H

#include "vec_sim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/debug_commands.c:1758: on success *neighborsData is an array of per-level arrays,
 * each prefixed by its length, NULL-terminated. */
int VecSimDebug_GetElementNeighborsInHNSWGraph(VecSimIndex *index, size_t label, int ***neighborsData);
/* reference src/debug_commands.c:1778 -- NULL-safe */
void VecSimDebug_ReleaseElementNeighborsInHNSWGraph(int **neighborsData);

#ifdef __cplusplus
}
#endif
#endif /* VECSIM_VEC_SIM_DEBUG_H */
