// corpus_kernels.hip -- keyed synthetic corpus, generated in place in HBM (gfx950).
//
// SURVEY.md 8(d) asks for a bench/test corpus any host can regenerate row by row: element (i, j) of the corpus is a
// pure function of (seed, i, j) -- Philox4x32-10 (Salmon et al., SC'11) keyed by the 64-bit seed, counter
// (i_lo, i_hi, j/4, 0), word j%4 -- so that the 30 GB bench corpus never crosses PCIe and still every returned
// neighbour can be re-scored from first principles on the host (oracle/flat_oracle.c oracle_philox_rows is the CPU
// twin; tests/test_gpu_philox.py compares them bit for bit, tests/test_oracle_flat.py pins the generator on the
// Random123 known answers).
//
// One thread makes one Philox block = 4 consecutive elements of one row and stores them with a single 4/8/16/32-byte
// write; rows are `stride` bytes apart, the padding behind `dim` is written as zeros (the scan kernels read whole
// 16-byte chunks).  Pure streaming writes: HBM-write bound.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"

namespace rsgpu {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// 24 random bits -> an fp32 in [-1, 1) on a 2^-23 grid: exact in fp32, so host and device agree bit for bit
__device__ __forceinline__ float unit_float(uint32_t w) { return (float)(w >> 8) * 0x1p-23f - 1.0f; }

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <int TYPE>
__global__ __launch_bounds__(256) void philox_rows_kernel(uint8_t *rows, size_t stride, uint32_t dim, uint32_t quads,
                                                          uint64_t seed, uint64_t first_index, uint32_t row_begin,
                                                          uint32_t n_rows) {
  const uint64_t total = (uint64_t)n_rows * quads;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)(t / quads), q = (uint32_t)(t % quads);
    const uint64_t gi = first_index + r;
    uint32_t w[4];
    philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), q, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = (4 * q + e < dim) ? unit_float(w[e]) : 0.0f;
    uint8_t *dst = rows + (size_t)(row_begin + r) * stride;
    if constexpr (TYPE == KT_F32) {
      *reinterpret_cast<float4 *>(dst + (size_t)q * 16) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (TYPE == KT_F64) {
      double2 *d = reinterpret_cast<double2 *>(dst + (size_t)q * 32);
      d[0] = make_double2((double)v[0], (double)v[1]);
      if ((size_t)q * 32 + 16 < stride) d[1] = make_double2((double)v[2], (double)v[3]);  // odd dims: stride % 32 == 16
    } else if constexpr (TYPE == KT_F16) {
      ushort4 o;
      o.x = __half_as_ushort(__float2half_rn(v[0]));
      o.y = __half_as_ushort(__float2half_rn(v[1]));
      o.z = __half_as_ushort(__float2half_rn(v[2]));
      o.w = __half_as_ushort(__float2half_rn(v[3]));
      *reinterpret_cast<ushort4 *>(dst + (size_t)q * 8) = o;
    } else if constexpr (TYPE == KT_BF16) {
      ushort4 o;
      o.x = f32_to_bf16_rne(v[0]);
      o.y = f32_to_bf16_rne(v[1]);
      o.z = f32_to_bf16_rne(v[2]);
      o.w = f32_to_bf16_rne(v[3]);
      *reinterpret_cast<ushort4 *>(dst + (size_t)q * 8) = o;
    } else {  // INT8 / UINT8: the top byte of each word (as int8: the same bits, two's complement)
      uint32_t o = 0;
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (4 * q + e < dim) o |= (w[e] >> 24) << (8 * e);
      *reinterpret_cast<uint32_t *>(dst + (size_t)q * 4) = o;
    }
  }
}

void launch_philox_rows(void *rows, size_t stride, uint32_t dim, int type, uint64_t seed, uint64_t first_index,
                        uint32_t row_begin, uint32_t n_rows, hipStream_t s) {
  if (!n_rows) return;
  const size_t esz = type == KT_F64 ? 8 : (type == KT_F32 ? 4 : (type == KT_F16 || type == KT_BF16 ? 2 : 1));
  const uint32_t quads = (uint32_t)((stride + 4 * esz - 1) / (4 * esz));  // stride is a multiple of 16 (f64: maybe not of 32)
  const uint64_t total = (uint64_t)n_rows * quads;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 32u);
  uint8_t *p = static_cast<uint8_t *>(rows);
  switch (type) {
    case KT_F32: hipLaunchKernelGGL(philox_rows_kernel<KT_F32>, grid, 256, 0, s, p, stride, dim, quads, seed, first_index, row_begin, n_rows); break;
    case KT_F64: hipLaunchKernelGGL(philox_rows_kernel<KT_F64>, grid, 256, 0, s, p, stride, dim, quads, seed, first_index, row_begin, n_rows); break;
    case KT_F16: hipLaunchKernelGGL(philox_rows_kernel<KT_F16>, grid, 256, 0, s, p, stride, dim, quads, seed, first_index, row_begin, n_rows); break;
    case KT_BF16: hipLaunchKernelGGL(philox_rows_kernel<KT_BF16>, grid, 256, 0, s, p, stride, dim, quads, seed, first_index, row_begin, n_rows); break;
    default: hipLaunchKernelGGL(philox_rows_kernel<KT_I8>, grid, 256, 0, s, p, stride, dim, quads, seed, first_index, row_begin, n_rows); break;
  }
}

// *out_bits = max(*out_bits, bits of v[i]) over v[begin, end): non-negative floats order like their bits
__global__ __launch_bounds__(256) void max_f32_bits_kernel(const float *__restrict__ v, uint32_t begin, uint32_t end, uint32_t *__restrict__ out_bits) {
  uint32_t m = 0;
  for (uint32_t i = begin + blockIdx.x * 256 + threadIdx.x; i < end; i += gridDim.x * 256) {
    const uint32_t b = __float_as_uint(v[i]);
    m = b > m && !(b & 0x80000000u) ? b : m;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out_bits, m);
}
void launch_max_f32_bits(const float *v, uint32_t begin, uint32_t end, uint32_t *out_bits, hipStream_t s) {
  if (end <= begin) return;
  const uint32_t need = (end - begin + 255) / 256;
  hipLaunchKernelGGL(max_f32_bits_kernel, dim3(need < 1024 ? need : 1024), dim3(256), 0, s, v, begin, end, out_bits);
}

// ---- label table maintenance (label_table.cpp) ----
__global__ __launch_bounds__(256) void label_fill_kernel(uint32_t *__restrict__ dst, size_t begin, size_t end, size_t lo, size_t hi,
                                                         uint32_t first) {
  for (size_t i = begin + (size_t)blockIdx.x * 256 + threadIdx.x; i < end; i += (size_t)gridDim.x * 256)
    dst[i] = (i >= lo && i < hi) ? first + (uint32_t)(i - lo) : kNoRow;
}
__global__ __launch_bounds__(256) void label_scatter_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ idx,
                                                            const uint32_t *__restrict__ val, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[idx[i]] = val[i];
}
__global__ __launch_bounds__(256) void label_decode_kernel(uint32_t *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] -= 1u;
}
void launch_label_fill(uint32_t *dst, size_t begin, size_t end, size_t lo, size_t hi, uint32_t first, hipStream_t s) {
  if (end <= begin) return;
  const size_t need = (end - begin + 255) / 256;
  hipLaunchKernelGGL(label_fill_kernel, dim3((uint32_t)std::min<size_t>(need, 256u * 32u)), dim3(256), 0, s, dst, begin, end, lo, hi, first);
}
void launch_label_scatter(uint32_t *dst, const uint32_t *idx, const uint32_t *val, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(label_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, idx, val, n);
}
__global__ __launch_bounds__(256) void label_scatter16_kernel(uint4 *__restrict__ dst, const uint32_t *__restrict__ idx,
                                                              const uint4 *__restrict__ val, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[idx[i]] = val[i];
}
void launch_label_scatter16(void *dst, const uint32_t *idx, const void *val, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(label_scatter16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (uint4 *)dst, idx, (const uint4 *)val, n);
}
void launch_label_decode(uint32_t *dst, size_t n, hipStream_t s) {
  if (!n) return;
  const size_t need = (n + 255) / 256;
  hipLaunchKernelGGL(label_decode_kernel, dim3((uint32_t)std::min<size_t>(need, 256u * 32u)), dim3(256), 0, s, dst, n);
}

}  // namespace rsgpu
