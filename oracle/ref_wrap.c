/* oracle/ref_wrap.c -- exports of the REFERENCE's own code for the few pieces of the path that compile from their
 * own sources without the Rust workspace or the un-vendored submodules (test infrastructure, like the rest of oracle/).
 * Built by `make -C oracle ref` from the sources where they lie under $(REF) (= /root/reference) into oracle/_ref/;
 * nothing of the reference is copied into this repository.
 *
 *   src/vector_normalization.h:37-92   VectorNorm_L2 / _IP / _Cosine, getVectorNormalizationFunction (static inline;
 *                                      compiled against THIS repository's include/VecSim/vec_sim_common.h, which it
 *                                      includes for VecSimMetric -- a drop-in check of that header as a side effect)
 *   src/util/minmax_heap.c             the min-max heap behind the hybrid iterator's top-K (hybrid_reader.c:88-138,
 *                                      446-470) -- compiled as its own object by the Makefile, not through this file
 */
#include "vector_normalization.h"

double ref_vector_norm(int metric, double value) {
  return getVectorNormalizationFunction((VecSimMetric)metric)(value);
}

/* src/hybrid/hybrid_scoring.c (HybridRRFScore / HybridLinearScore, :41-84) is compiled as its own object; its
 * EXPLAINSCORE formatters use hiredis' sds (an empty submodule).  They are not on this path: these stand-ins only
 * let the library load with RTLD_NOW. */
#ifdef REF_WRAP_SDS_STUBS
#include <stdlib.h>
char *sdsnew(const char *s) { (void)s; abort(); }
char *sdscat(char *s, const char *t) { (void)s; (void)t; abort(); }
char *sdscatprintf(char *s, const char *fmt, ...) { (void)s; (void)fmt; abort(); }
void sdsfree(char *s) { (void)s; abort(); }
#endif
