// exchange_kernels.hip -- the merge step of the multi-GPU exchange on the device (gfx950).
//
// After the RCCL all-gather every rank holds world x k candidates {u64 label, u64 orderable key of the fp64 score} (16 bytes
// each; label == UINT64_MAX marks padding).  The k best by (distance, label) ascending -- the order of the host merge
// RSGPU_MergeTopKHost and of the reference coordinator's heap (src/module.c:3541-3547) -- are selected by RANK: every
// candidate counts the candidates that precede it in the total order; ranks below k ARE the answer, written straight to
// their slot in pinned host memory.  n <= 8192 candidates live in LDS (128 KiB); n^2 / 1024 comparisons per thread is
// nothing at the sizes the exchange has (8 ranks x 10: 80 candidates).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace rsgpu {
namespace {

__global__ __launch_bounds__(1024) void merge_topk_kernel(const uint4 *__restrict__ all, uint32_t n, uint32_t k,
                                                          uint4 *__restrict__ out, uint32_t *__restrict__ out_n) {
  extern __shared__ uint4 cand[];
  for (uint32_t i = threadIdx.x; i < n; i += 1024) cand[i] = all[i];
  __shared__ uint32_t valid;
  if (threadIdx.x == 0) valid = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    const uint4 me = cand[i];
    const uint64_t my_label = ((uint64_t)me.y << 32) | me.x;
    const uint64_t my_key = ((uint64_t)me.w << 32) | me.z;
    if (my_label == ~0ull) continue;  // padding
    atomicAdd(&valid, 1u);
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) {
      const uint4 o = cand[j];
      const uint64_t ol = ((uint64_t)o.y << 32) | o.x;
      // (key, label, position): the position only separates duplicates of one (key, label) pair, which the exchange does
      // not produce -- shards hold disjoint labels -- but a rank must be a permutation whatever comes in
      const uint64_t ok = ((uint64_t)o.w << 32) | o.z;
      const bool before = ol != ~0ull && (ok < my_key || (ok == my_key && (ol < my_label || (ol == my_label && j < i))));
      rank += before ? 1u : 0u;
    }
    if (rank < k) out[rank] = me;
  }
  __syncthreads();
  if (threadIdx.x == 0) *out_n = valid < k ? valid : k;
}

}  // namespace

bool launch_merge_topk(const void *all, uint32_t n, uint32_t k, void *out_pinned, uint32_t *out_n_pinned, hipStream_t s) {
  if (!n || n > 8192) return false;
  // above the 64 KiB default the dynamic LDS of a kernel must be raised explicitly (per device; fusion_kernels.hip does the
  // same): without it launches with n > 4096 candidates fail.  Refused: the caller merges on the host.
  if ((size_t)n * 16 > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(merge_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
  }
  hipLaunchKernelGGL(merge_topk_kernel, dim3(1), dim3(1024), (size_t)n * 16, s, (const uint4 *)all, n, k, (uint4 *)out_pinned,
                     out_n_pinned);
  return true;
}

}  // namespace rsgpu
