// scan_ops.hpp -- device pieces shared by the FLAT scan kernels (scan_kernels.hip: one query per pass;
// scan_mq_kernels.hip: several queries per pass): the 16-byte chunk operators per element type and metric, the group
// reduction, the distance -> orderable key mapping and the row-shape table.  Both kernels MUST compute a row's distance
// with the same per-lane operation order and the same reduction tree: replies of a coalesced pass are bit-identical to
// single queries because these definitions are the only ones there are.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace rsgpu {
namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));  // one 16-byte chunk
__device__ __forceinline__ u4 zero4() { return (u4){0u, 0u, 0u, 0u}; }

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xFFFFFFFFu;  // NaN sorts last
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ... and back (0xFFFFFFFF, the key of every NaN, comes back as a NaN)
__device__ __forceinline__ float key2f(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

template <bool NT>
__device__ __forceinline__ u4 load16(const u4 *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// ---- per element type: accumulator, distance and key types ---------------------------------------
struct I2 {
  int xq, xx;  // sum x*q and sum x*x (bit patterns of u32 sums for KT_U8)
};
template <int TYPE>
struct Tr {
  typedef float acc_t;
  typedef float out_t;
  typedef uint32_t key_t;
  static constexpr bool kExtra = false;  // the query carries one more chunk {sum q^2, |q|}
};
template <>
struct Tr<KT_F64> {
  typedef double acc_t;
  typedef double out_t;
  typedef uint64_t key_t;
  static constexpr bool kExtra = false;
};
template <>
struct Tr<KT_I8> {
  typedef I2 acc_t;
  typedef float out_t;
  typedef uint32_t key_t;
  static constexpr bool kExtra = true;
};
template <>
struct Tr<KT_U8> : Tr<KT_I8> {};

__device__ __forceinline__ uint64_t d2key(double f) {
  uint64_t u = (uint64_t)__double_as_longlong(f);
  if ((u & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) return ~0ull;  // NaN sorts last
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ uint32_t to_key(float f) { return f2key(f); }
__device__ __forceinline__ uint64_t to_key(double f) { return d2key(f); }
__device__ __forceinline__ void set_nan(float &f) { f = __uint_as_float(0x7fc00000u); }
__device__ __forceinline__ void set_nan(double &f) { f = __longlong_as_double(0x7ff8000000000000ll); }

// ---- one 16-byte chunk of a row against the matching query chunk ---------------------------------
template <int TYPE, int METRIC>
struct Op;

template <>
struct Op<KT_F32, KM_IP> {
  static __device__ __forceinline__ float add(float acc, u4 x, u4 q) {
    acc = fmaf(__uint_as_float(x.x), __uint_as_float(q.x), acc);
    acc = fmaf(__uint_as_float(x.y), __uint_as_float(q.y), acc);
    acc = fmaf(__uint_as_float(x.z), __uint_as_float(q.z), acc);
    acc = fmaf(__uint_as_float(x.w), __uint_as_float(q.w), acc);
    return acc;
  }
};
template <>
struct Op<KT_F32, KM_L2> {
  static __device__ __forceinline__ float add(float acc, u4 x, u4 q) {
    float d0 = __uint_as_float(x.x) - __uint_as_float(q.x);
    float d1 = __uint_as_float(x.y) - __uint_as_float(q.y);
    float d2 = __uint_as_float(x.z) - __uint_as_float(q.z);
    float d3 = __uint_as_float(x.w) - __uint_as_float(q.w);
    acc = fmaf(d0, d0, acc);
    acc = fmaf(d1, d1, acc);
    acc = fmaf(d2, d2, acc);
    acc = fmaf(d3, d3, acc);
    return acc;
  }
};
// fp64: two elements per chunk, fp64 accumulate
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double as_d(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
template <>
struct Op<KT_F64, KM_IP> {
  static __device__ __forceinline__ double add(double acc, u4 x, u4 q) {
    // (one 16-byte value -> two doubles in ONE bit cast: building each double from two words let the compiler narrow
    // the chunk load into a dwordx4 + a second dwordx2 of the same address and wait for every load separately)
    const d2 xv = __builtin_bit_cast(d2, x), qv = __builtin_bit_cast(d2, q);
    acc = fma(xv.x, qv.x, acc);
    return fma(xv.y, qv.y, acc);
  }
};
template <>
struct Op<KT_F64, KM_L2> {
  static __device__ __forceinline__ double add(double acc, u4 x, u4 q) {
    const d2 xv = __builtin_bit_cast(d2, x), qv = __builtin_bit_cast(d2, q);
    double d0 = xv.x - qv.x, d1 = xv.y - qv.y;
    acc = fma(d0, d0, acc);
    return fma(d1, d1, acc);
  }
};
// fp16: products of two halves are exact in fp32, accumulation is fp32 (v_dot2_f32_f16)
__device__ __forceinline__ half2_t as_h2(uint32_t u) {
  half2_t h;
  __builtin_memcpy(&h, &u, 4);
  return h;
}
template <>
struct Op<KT_F16, KM_IP> {
  static __device__ __forceinline__ float add(float acc, u4 x, u4 q) {
    acc = __builtin_amdgcn_fdot2(as_h2(x.x), as_h2(q.x), acc, false);
    acc = __builtin_amdgcn_fdot2(as_h2(x.y), as_h2(q.y), acc, false);
    acc = __builtin_amdgcn_fdot2(as_h2(x.z), as_h2(q.z), acc, false);
    acc = __builtin_amdgcn_fdot2(as_h2(x.w), as_h2(q.w), acc, false);
    return acc;
  }
};
__device__ __forceinline__ float l2_h2(float acc, uint32_t a, uint32_t b) {
  half2_t x = as_h2(a), y = as_h2(b);
  float d0 = (float)x.x - (float)y.x, d1 = (float)x.y - (float)y.y;
  acc = fmaf(d0, d0, acc);
  return fmaf(d1, d1, acc);
}
template <>
struct Op<KT_F16, KM_L2> {
  static __device__ __forceinline__ float add(float acc, u4 x, u4 q) {
    acc = l2_h2(acc, x.x, q.x);
    acc = l2_h2(acc, x.y, q.y);
    acc = l2_h2(acc, x.z, q.z);
    return l2_h2(acc, x.w, q.w);
  }
};
// bf16: widen by shifting into the top half of an fp32
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float ip_bf2(float acc, uint32_t a, uint32_t b) {
  acc = fmaf(bf_lo(a), bf_lo(b), acc);
  return fmaf(bf_hi(a), bf_hi(b), acc);
}
__device__ __forceinline__ float l2_bf2(float acc, uint32_t a, uint32_t b) {
  float d0 = bf_lo(a) - bf_lo(b), d1 = bf_hi(a) - bf_hi(b);
  acc = fmaf(d0, d0, acc);
  return fmaf(d1, d1, acc);
}
template <>
struct Op<KT_BF16, KM_IP> {
  static __device__ __forceinline__ float add(float acc, u4 x, u4 q) {
    acc = ip_bf2(acc, x.x, q.x);
    acc = ip_bf2(acc, x.y, q.y);
    acc = ip_bf2(acc, x.z, q.z);
    return ip_bf2(acc, x.w, q.w);
  }
};
template <>
struct Op<KT_BF16, KM_L2> {
  static __device__ __forceinline__ float add(float acc, u4 x, u4 q) {
    acc = l2_bf2(acc, x.x, q.x);
    acc = l2_bf2(acc, x.y, q.y);
    acc = l2_bf2(acc, x.z, q.z);
    return l2_bf2(acc, x.w, q.w);
  }
};
// int8 / uint8: exact integer sums on the packed dot-product units (v_dot4_i32_i8 / v_dot4_u32_u8),
// 16 elements per chunk.  L2 = sum x^2 + sum q^2 - 2 sum xq needs sum x^2 next to the dot, cosine needs
// it for |x|; plain IP does not.
template <int TYPE>
__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) {
  if (TYPE == KT_I8) return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
  return (int)__builtin_amdgcn_udot4(a, b, (uint32_t)c, false);
}
template <int TYPE, int METRIC>
struct OpInt {
  static __device__ __forceinline__ I2 add(I2 acc, u4 x, u4 q) {
    acc.xq = dot4<TYPE>(x.x, q.x, acc.xq);
    acc.xq = dot4<TYPE>(x.y, q.y, acc.xq);
    acc.xq = dot4<TYPE>(x.z, q.z, acc.xq);
    acc.xq = dot4<TYPE>(x.w, q.w, acc.xq);
    if (METRIC == KM_L2 || METRIC == KM_COS) {
      acc.xx = dot4<TYPE>(x.x, x.x, acc.xx);
      acc.xx = dot4<TYPE>(x.y, x.y, acc.xx);
      acc.xx = dot4<TYPE>(x.z, x.z, acc.xx);
      acc.xx = dot4<TYPE>(x.w, x.w, acc.xx);
    }
    return acc;
  }
};
template <int METRIC>
struct Op<KT_I8, METRIC> : OpInt<KT_I8, METRIC> {};
template <int METRIC>
struct Op<KT_U8, METRIC> : OpInt<KT_U8, METRIC> {};

template <int G>
__device__ __forceinline__ float group_reduce(float v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
template <int G>
__device__ __forceinline__ double group_reduce(double v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
template <int G>
__device__ __forceinline__ I2 group_reduce(I2 v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) {
    v.xq += __shfl_xor(v.xq, m, 64);
    v.xx += __shfl_xor(v.xx, m, 64);
  }
  return v;
}

// reduced sums -> distance.  qx: the extra query chunk of the integer types {sum q^2, |q| as f32 bits}
template <int TYPE, int METRIC>
__device__ __forceinline__ float finish(float acc, u4) {
  return METRIC == KM_IP ? 1.0f - acc : acc;
}
template <int TYPE, int METRIC>
__device__ __forceinline__ double finish(double acc, u4) {
  return METRIC == KM_IP ? 1.0 - acc : acc;
}
template <int TYPE, int METRIC>
__device__ __forceinline__ float finish(I2 acc, u4 qx) {
  // the integer value is exact (|.| < 2^63); it becomes a float once, as in `float(res)` of the scalar loop
  const long long xq = TYPE == KT_I8 ? (long long)acc.xq : (long long)(uint32_t)acc.xq;
  const long long xx = TYPE == KT_I8 ? (long long)acc.xx : (long long)(uint32_t)acc.xx;
  const long long qq = TYPE == KT_I8 ? (long long)(int)qx.x : (long long)qx.x;
  if (METRIC == KM_L2) return (float)(xx + qq - 2 * xq);
  if (METRIC == KM_IP) return 1.0f - (float)xq;
  if (METRIC == KM_IPS || METRIC == KM_L2S) return (float)xq;  // scaled per row where the key is stored
  return 1.0f - (float)xq / (sqrtf((float)xx) * __uint_as_float(qx.y));
}

// int8-shadow distance of a row from its integer dot: meta = {row scale, |x|^2}, qx = {0, query scale, |q|^2, 0}
template <int METRIC>
__device__ __forceinline__ float shadow8_distance(float dot, float2 meta, u4 qx) {
  const float xq = dot * (meta.x * __uint_as_float(qx.y));
  return METRIC == KM_L2S ? (__uint_as_float(qx.z) + meta.y) - 2.0f * xq : 1.0f - xq;
}


struct Shape {
  int G, ITERS;
};
inline Shape pick_shape(uint32_t chunks) {
  // 96 / 160 / 224 chunks (e.g. 768 halves): half a wavefront per row keeps every lane of every load busy,
  // a full one would idle 32 lanes in its last pass
  if (chunks > 64 && chunks % 64 == 32 && chunks <= 224) return {32, (int)(chunks / 32)};
  if (chunks == 48 || chunks == 80 || chunks == 112) return {16, (int)(chunks / 16)};  // e.g. 768 int8 elements
  if (chunks > 64) return {64, (int)((chunks + 63) / 64)};
  int g = 1;
  while ((uint32_t)g < chunks) g <<= 1;
  return {g, 1};
}

}  // namespace
}  // namespace rsgpu
