// fusion_kernels.hip -- FT.HYBRID fusion epilogue on the device (gfx950), the step right after the hot path:
// the ranked search list (BM25/TF-IDF scores) and the ranked vector list (KNN distances) become one list.
//
// Reference: RPHybridMerger (src/result_processor.c:2549-2571, 2613-2670: at most `window` results from each
// upstream, in upstream order), HybridRRFScore / HybridLinearScore (src/hybrid/hybrid_scoring.c:41-84),
// VectorNorm_L2 / _IP / _Cosine (src/vector_normalization.h:37-60), final order cmpByScore
// (src/result_processor.c:834-850: score descending, lower doc id first).
//
// The lists are tiny next to the scan (window is 20 by default), so this is ONE workgroup: fp64 arithmetic in
// exactly the reference's order (contribution of upstream 0, then upstream 1; compiled with
// -ffp-contract=off), id matching by a sweep over the other list held in LDS, and an exact rank by counting
// ((score, id) is a total order) instead of a sort.  Its point is that the two ranked lists can stay on the
// device between the scoring / KNN kernels and the reply.
#include <hip/hip_runtime.h>

#include "search_kernels.hpp"

namespace rsgpu {
namespace {

__device__ __forceinline__ double vector_norm(int metric, double d) {
  if (metric == 0) return 1.0 / (1.0 + d);         // L2
  if (metric == 1) return (1.0 + d) / 2.0;          // IP
  if (metric == 2) return (1.0 + (1.0 - d)) / 2.0;  // cosine
  return d;
}

// better(x, y): x ranks before y
__device__ __forceinline__ bool better(double sx, uint64_t ix, double sy, uint64_t iy) {
  return sx > sy || (sx == sy && ix < iy);
}

__global__ __launch_bounds__(1024) void hybrid_fuse_kernel(FuseParams p) {
  extern __shared__ unsigned char dyn[];
  // LDS: ids[na+nb] u64, score[na+nb] f64, valid[na+nb] u8
  const uint32_t na = p.na, nb = p.nb, m_all = na + nb;
  uint64_t *ids = reinterpret_cast<uint64_t *>(dyn);
  double *sc = reinterpret_cast<double *>(dyn + (size_t)m_all * 8);
  unsigned char *valid = dyn + (size_t)m_all * 16;
  __shared__ uint32_t n_valid;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) n_valid = 0;
  for (uint32_t i = tid; i < m_all; i += blockDim.x) {
    const bool a = i < na;
    const uint32_t j = a ? i : i - na;
    ids[i] = a ? p.a_ids[j] : p.b_ids[j];
    double c;
    if (p.scoring == 0) c = 1.0 / (p.constant + (double)(j + 1));
    else c = a ? p.w0 * p.a_scores[j] : p.w1 * vector_norm(p.metric, p.b_scores[j]);
    sc[i] = 0.0 + c;
    valid[i] = 1;
  }
  __syncthreads();
  // a document of list a that is also in list b takes b's contribution (added second) and retires b's entry;
  // the first match wins, as in the merger's dictionary
  for (uint32_t i = tid; i < na; i += blockDim.x) {
    const uint64_t id = ids[i];
    for (uint32_t j = 0; j < nb; j++)
      if (ids[na + j] == id) {
        sc[i] += sc[na + j];
        valid[na + j] = 0;
        break;
      }
  }
  __syncthreads();
  for (uint32_t i = tid; i < m_all; i += blockDim.x) {
    if (!valid[i]) continue;
    atomicAdd(&n_valid, 1u);
    const double s = sc[i];
    const uint64_t id = ids[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < m_all; j++) rank += (valid[j] && better(sc[j], ids[j], s, id)) ? 1u : 0u;
    p.ids_out[rank] = id;
    p.scores_out[rank] = s;
  }
  __syncthreads();
  if (tid == 0) p.count_out[0] = n_valid;
}

}  // namespace

void launch_hybrid_fuse(const FuseParams &p, hipStream_t s) {
  const size_t lds = (size_t)(p.na + p.nb) * 17;
  // up to 2 * 4096 entries = 136 KiB of the CU's 160 KiB: above the 64 KiB default cap
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(hybrid_fuse_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(2 * kFuseMaxWindow * 17));
  (void)attr;
  hipLaunchKernelGGL(hybrid_fuse_kernel, dim3(1), dim3(1024), lds, s, p);
}

}  // namespace rsgpu
