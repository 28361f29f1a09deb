// h8_quant.hpp -- the ONE definition of the in-flight fp16 -> int8 quantiser of the batched matrix-core pass over FLOAT16 rows
// (gemm_qs_f32_kernel<.., SRC_H8>, round 6).  No int8 copy of the corpus is stored: the pass reads the fp16 rows the exact scan
// reads and quantises them on their way from LDS to the int8 matrix pipe; the error band of the filter comes from index-wide
// maxima (|x8|^2, |ex|^2) that h8_stats_kernel computes WITH THIS FUNCTION, so the band describes exactly what the pass sees.
//   q(x) = low byte of the fp16 bits of fma(x, inv, 1536):  fp16 values in [1024, 2048) have ulp 1, so the fused multiply-add
//   rounds x * inv (exact in the fma) to the nearest-even integer v in [-127, 127] and stores 512 + v in the mantissa -- whose
//   low eight bits are v in two's complement (512 = 0x200).  One v_pk_fma_f16 per two elements, one v_perm_b32 per four.
// inv = the largest fp16 <= 127 / max |x_i| (capped at 65504): |x * inv| <= 127, no clamp needed.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rsgpu {

typedef _Float16 h8_half2 __attribute__((ext_vector_type(2)));

// two fp16 elements (one dword) -> fp16 pair whose low bytes (bytes 0 and 2) are the int8 values
__device__ __forceinline__ uint32_t h8_quant2(uint32_t x2, uint32_t inv2) {
  h8_half2 x, s, m;
  const uint32_t magic = 0x66006600u;  // 1536.0h twice
  __builtin_memcpy(&x, &x2, 4);
  __builtin_memcpy(&s, &inv2, 4);
  __builtin_memcpy(&m, &magic, 4);
  const h8_half2 t = __builtin_elementwise_fma(x, s, m);
  uint32_t r;
  __builtin_memcpy(&r, &t, 4);
  return r;
}
// bytes 0 and 2 of lo, bytes 0 and 2 of hi -> one dword of four int8 (element order kept)
__device__ __forceinline__ uint32_t h8_pack4(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }

// ---- the same for FLOAT32 rows (gemm_qs_h8r_kernel<.., SRC_F8>): q(x) = low byte of the fp32 bits of fma(x, inv, 1.5 * 2^23) -- fp32
// values in [2^23, 2^24) have ulp 1, the fused multiply-add rounds x * inv to the nearest-even integer v in [-127, 127] and the
// mantissa's low byte is v in two's complement.  inv = the largest fp32 <= 127 / max |x_i|.  Four elements (one 16-byte chunk)
// -> one dword of four int8: four v_fma_f32 + three v_perm_b32.
typedef uint32_t h8_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t f8_quant1(uint32_t x, float inv) { return __float_as_uint(__builtin_fmaf(__uint_as_float(x), inv, 12582912.0f)); }
__device__ __forceinline__ uint32_t f8_quant4(const h8_u4 &x, float inv) {
  const uint32_t t0 = f8_quant1(x[0], inv), t1 = f8_quant1(x[1], inv), t2 = f8_quant1(x[2], inv), t3 = f8_quant1(x[3], inv);
  const uint32_t lo = __builtin_amdgcn_perm(t1, t0, 0x0c0c0400u);  // {t0.b0, t1.b0, 0, 0}
  const uint32_t hi = __builtin_amdgcn_perm(t3, t2, 0x04000c0cu);  // {0, 0, t2.b0, t3.b0}
  return lo | hi;
}

}  // namespace rsgpu
