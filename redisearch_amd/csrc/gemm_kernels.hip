// gemm_kernels.hip -- batched-query FLAT KNN on the matrix cores (gfx950 MFMA), hand-written HIP.
//
// The reference has no batched API: B queries are B independent VecSimIndex_TopKQuery calls
// (reference src/iterators/hybrid_reader.c:374), each streaming the whole corpus.  With B = 256 fp16
// queries the same corpus pass becomes a true GEMM  S[256 x N] = Q[256 x dim] * X^T  with arithmetic
// intensity 256 flop/B -- at the MFMA/HBM ridge of MI355X -- so THIS is where MFMA belongs
// (BASELINE configs[2]; the single-query path in scan_kernels.hip stays a bandwidth-bound GEMV).
//
// S is never materialised (256 x 10M fp32 = 10 GB).  Three steps, all on the device:
//   1. the GEMM kernel in KEYS mode over a small prefix of the corpus (n0 rows) writes every
//      orderable distance key; batch_threshold_kernel finds, per query, the k-th smallest key of the
//      sample: a valid upper bound tau[q] of the final k-th distance;
//   2. the GEMM kernel in FILTER mode over ALL rows keeps only (row,key) with distance <= tau[q]
//      -- expected k*N/n0 survivors per query -- appended to per-query candidate lists;
//   3. batch_final_select_kernel picks, per query, the exact k smallest (key,row) composites of its
//      candidates (same total order as select_kernels.hip).
// The result is exact whatever the data order (tau is a true upper bound); a candidate overflow makes
// the host fall back to the single-query path for that query.
//
// GEMM kernel: 512 threads = 8 wavefronts (4 along the queries x 2 along the corpus rows), block tile
// 256 queries x 128 rows x BK=64, v_mfma_f32_32x32x16_f16 (fp32 accumulate), each wave owns a
// 64 x 64 sub-tile = 2 x 2 MFMA tiles = 64 accumulator VGPRs (a 256 x 256 tile with 128 accumulators
// per lane spilled ~100 VGPRs under hipcc, so the smaller tile is the faster one here).  Both operands
// are K-contiguous rows (an "NT" GEMM), staged global -> registers -> LDS in full 128-byte lines
// (8 lanes x 16 B per row), double-buffered (96 KiB of the CU's 160 KiB LDS), 16-byte chunks
// XOR-swizzled by (row & 7) so the ds_read_b128 fragment reads are at most 2-way conflicted.
// Persistent grid: one workgroup per CU walks the corpus tiles; the 384 KiB query matrix is re-read
// from L2 per tile.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace rsgpu {
namespace {

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int BM = 256, BN = 128, BKC = 8;  // BKC: 16-byte chunks per K step (64 halves)
constexpr int A_CHUNKS = BM * BKC;          // 2048 chunks = 32 KiB of queries per stage
constexpr int B_CHUNKS = BN * BKC;          // 1024 chunks = 16 KiB of corpus rows per stage
constexpr int STAGE_CHUNKS = A_CHUNKS + B_CHUNKS;

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xFFFFFFFFu;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int DT>
__device__ __forceinline__ f32x16 mfma(u4 a, u4 b, f32x16 c) {
  if (DT == KT_F16) {
    half8 x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
  } else {
    bf16x8 x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  }
}

// swizzled chunk slot of (row, chunk) inside a [256][8] tile
__device__ __forceinline__ uint32_t slot(uint32_t row, uint32_t c) { return row * BKC + (c ^ (row & 7u)); }

struct GemmArgs {
  const u4 *rows;      // corpus, row-contiguous
  const u4 *queries;   // [256][stride16], zero padded
  uint32_t stride16;   // row stride in 16-byte chunks == K chunks
  uint32_t row_begin, row_end;
  int mode;            // 0: write keys, 1: filter
  uint32_t *keys_out;  // mode 0: [256][keys_ld], column = row - row_begin
  uint32_t keys_ld;
  const float *tau;    // mode 1: per-query distance upper bound
  uint32_t *cand_count;  // [256]
  uint2 *cand;           // [256][cand_cap] (row, key)
  uint32_t cand_cap;
};

template <int DT>
__global__ __launch_bounds__(512) void gemm_topk_kernel(GemmArgs g) {
  __shared__ u4 smem[2 * STAGE_CHUNKS];  // [stage][A 256x8 | B 128x8] = 96 KiB
  __shared__ float tau_s[BM];
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t wm = w >> 1, wn = w & 1;  // wave position: 4 along the queries x 2 along the rows
  const uint32_t kchunks = g.stride16;
  const uint32_t ksteps = (kchunks + BKC - 1) / BKC;
  const uint32_t n = g.row_end - g.row_begin;
  const uint32_t n_tiles = (n + BN - 1) / BN;
  if (g.mode == 1) {
    if (tid < BM) tau_s[tid] = g.tau[tid];
  }
  __syncthreads();

  // staging map: this thread moves chunks id = tid + i*512 of the query tile (i<4) and of the corpus
  // tile (i<2): row id/8, chunk id%8 -- 8 consecutive lanes cover one 128-byte line
  const uint32_t schunk = tid & 7u, srow0 = tid >> 3;  // rows srow0 + 64*i

  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t row0 = g.row_begin + tile * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.0f;

    u4 ra[4], rb[2];
    auto load_stage = [&](uint32_t ks) {
      const uint32_t c = ks * BKC + schunk;
      const bool in_k = c < kchunks;
#pragma unroll
      for (int i = 0; i < 4; i++)
        ra[i] = in_k ? g.queries[(size_t)(srow0 + 64 * i) * kchunks + c] : (u4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 2; i++) {
        uint32_t xr = row0 + srow0 + 64 * i;
        if (xr >= g.row_end) xr = g.row_end - 1;  // tail tile: recomputed, never emitted
        rb[i] = in_k ? __builtin_nontemporal_load(g.rows + (size_t)xr * kchunks + c) : (u4){0u, 0u, 0u, 0u};
      }
    };
    auto store_stage = [&](uint32_t buf) {
      u4 *A = smem + buf * STAGE_CHUNKS, *B = A + A_CHUNKS;
#pragma unroll
      for (int i = 0; i < 4; i++) A[slot(srow0 + 64 * i, schunk)] = ra[i];
#pragma unroll
      for (int i = 0; i < 2; i++) B[slot(srow0 + 64 * i, schunk)] = rb[i];
    };

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (uint32_t ks = 0; ks < ksteps; ks++) {
      const uint32_t buf = ks & 1u;
      if (ks + 1 < ksteps) load_stage(ks + 1);  // in flight during the MFMAs below
      const u4 *A = smem + buf * STAGE_CHUNKS, *B = A + A_CHUNKS;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const uint32_t c = kk * 2 + (lane >> 5);
        u4 a[2], b[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) a[mt] = A[slot(wm * 64 + mt * 32 + (lane & 31), c)];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) b[nt] = B[slot(wn * 64 + nt * 32 + (lane & 31), c)];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) acc[mt][nt] = mfma<DT>(a[mt], b[nt], acc[mt][nt]);
      }
      if (ks + 1 < ksteps) store_stage(buf ^ 1u);
      __syncthreads();
    }

    // epilogue: C row = query (reg&3)+8*(reg>>2)+4*(lane>>5) of the MFMA tile, C col = corpus row lane&31
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const uint32_t xr = row0 + wn * 64 + nt * 32 + (lane & 31);
        const bool live = xr < g.row_end;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const uint32_t q = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float d = 1.0f - acc[mt][nt][r];
          if (g.mode == 0) {
            if (live) g.keys_out[(size_t)q * g.keys_ld + (xr - g.row_begin)] = f2key(d);
          } else if (live && d <= tau_s[q]) {
            uint32_t s = atomicAdd(&g.cand_count[q], 1u);
            if (s < g.cand_cap) g.cand[(size_t)q * g.cand_cap + s] = make_uint2(xr, f2key(d));
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---- per-query block-level radix select ---------------------------------------------------------------
// One workgroup per query.  PAIRS=false: elements are keys[q*ld + i], i<n, composite (key,i);
// PAIRS=true: elements are cand[q*cap + i] = (row,key), i<min(count[q],cap), composite (key,row).
// Finds the k smallest composites; writes the k-th smallest key as a float threshold (tau_out) and/or
// the winners (unordered) to out_rows/out_keys[q*k_ld + j], out_n[q] = how many.
struct BatchSel {
  const uint32_t *keys;
  uint32_t ld, n;
  const uint2 *cand;
  const uint32_t *cand_count;
  uint32_t cand_cap;
  uint32_t k;
  float *tau_out;
  uint32_t *out_rows, *out_keys, *out_n;
  uint32_t k_ld;
  uint32_t *overflow;  // [q] set when a candidate list overflowed
  uint32_t n_valid;    // queries >= n_valid are padding: tau = -inf so that they never collect candidates
};

template <bool PAIRS>
__global__ __launch_bounds__(256) void batch_select_kernel(BatchSel s) {
  __shared__ uint32_t hist[256];
  __shared__ u64 sh_prefix;
  __shared__ uint32_t sh_krem, sh_exact, sh_levels, sh_out;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  uint32_t n = s.n;
  if (PAIRS) {
    uint32_t c = s.cand_count[q];
    if (c > s.cand_cap) {
      if (tid == 0) s.overflow[q] = 1;
      c = s.cand_cap;
    }
    n = c;
  }
  const uint32_t k = s.k < n ? s.k : n;
  if (tid == 0) {
    sh_prefix = 0;
    sh_krem = k;
    sh_exact = (k == 0 || k == n) ? 1u : 0u;  // everything (or nothing) is selected
    sh_levels = 0;
    sh_out = 0;
  }
  __syncthreads();
  auto elem = [&](uint32_t i) -> u64 {
    if (PAIRS) {
      uint2 e = s.cand[(size_t)q * s.cand_cap + i];
      return ((u64)e.y << 32) | e.x;
    }
    return ((u64)s.keys[(size_t)q * s.ld + i] << 32) | i;
  };
  const bool take_all = (k == n);
  for (int p = 0; p < 8 && !sh_exact; p++) {
    hist[tid] = 0;
    __syncthreads();
    const u64 prefix = sh_prefix;
    const int mshift = 64 - 8 * p, dshift = 56 - 8 * p;
    for (uint32_t i = tid; i < n; i += 256) {
      u64 c = elem(i);
      if (p == 0 || (c >> mshift) == (prefix >> mshift)) atomicAdd(&hist[(uint32_t)(c >> dshift) & 0xffu], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // wavefront 0 picks the digit
      uint32_t h0 = hist[lane * 4], h1 = hist[lane * 4 + 1], h2 = hist[lane * 4 + 2], h3 = hist[lane * 4 + 3];
      uint32_t sum = h0 + h1 + h2 + h3, inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += t;
      }
      const uint32_t exc = inc - sum, k_rem = sh_krem;
      const bool mine = exc < k_rem && k_rem <= inc;
      if (mine) {
        uint32_t c = exc, D, below, cnt;
        if (k_rem <= c + h0) { D = 0; below = c; cnt = h0; }
        else {
          c += h0;
          if (k_rem <= c + h1) { D = 1; below = c; cnt = h1; }
          else {
            c += h1;
            if (k_rem <= c + h2) { D = 2; below = c; cnt = h2; }
            else { c += h2; D = 3; below = c; cnt = h3; }
          }
        }
        D += lane * 4;
        sh_prefix = prefix | ((u64)D << dshift);
        sh_krem = k_rem - below;
        sh_levels = p + 1;
        if (cnt == k_rem - below) sh_exact = 1;
      }
    }
    __syncthreads();
  }
  // inclusive upper bound of the selection
  u64 hi = ~0ull;
  if (!take_all && k > 0) {
    const int sh = 64 - 8 * (int)sh_levels;
    hi = sh_prefix | (sh > 0 && sh < 64 ? ((1ull << sh) - 1ull) : 0ull);
  }
  if (tid == 0 && s.tau_out) {
    // k-th smallest key as a distance: the selection's bound, key part (ties included by the filter)
    uint32_t kk = (k == 0) ? 0u : (uint32_t)(hi >> 32);
    if (q >= s.n_valid) {
      s.tau_out[q] = __uint_as_float(0xff800000u);
    } else if (take_all) {  // fewer than k elements in the sample: no bound
      s.tau_out[q] = __uint_as_float(0x7f800000u);
    } else {
      uint32_t u = (kk & 0x80000000u) ? (kk ^ 0x80000000u) : ~kk;
      s.tau_out[q] = __uint_as_float(u);
    }
  }
  if (s.out_rows) {
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
      const uint32_t i = i0 + tid;
      bool take = false;
      u64 c = 0;
      if (i < n && k > 0) {
        c = elem(i);
        take = c <= hi;
      }
      if (take) {
        uint32_t o = atomicAdd(&sh_out, 1u);
        if (o < s.k_ld) {
          s.out_rows[(size_t)q * s.k_ld + o] = (uint32_t)c;
          s.out_keys[(size_t)q * s.k_ld + o] = (uint32_t)(c >> 32);
        }
      }
    }
    __syncthreads();
    if (tid == 0) s.out_n[q] = sh_out < s.k_ld ? sh_out : s.k_ld;
  }
}

}  // namespace

void launch_gemm_topk(int dtype, const void *rows, const void *queries, uint32_t stride16, uint32_t row_begin,
                      uint32_t row_end, int mode, uint32_t *keys_out, uint32_t keys_ld, const float *tau,
                      uint32_t *cand_count, void *cand, uint32_t cand_cap, hipStream_t s) {
  if (row_end <= row_begin) return;
  GemmArgs g{(const u4 *)rows, (const u4 *)queries, stride16, row_begin, row_end, mode, keys_out, keys_ld, tau,
             cand_count, (uint2 *)cand, cand_cap};
  uint32_t n_tiles = (row_end - row_begin + BN - 1) / BN;
  uint32_t cus = (uint32_t)scan_tuning().num_cus;
  uint32_t grid = n_tiles < cus ? n_tiles : cus;
  if (dtype == KT_F16) hipLaunchKernelGGL(gemm_topk_kernel<KT_F16>, dim3(grid), dim3(512), 0, s, g);
  else hipLaunchKernelGGL(gemm_topk_kernel<KT_BF16>, dim3(grid), dim3(512), 0, s, g);
}

void launch_batch_threshold(const uint32_t *keys, uint32_t ld, uint32_t n, uint32_t k, uint32_t n_queries,
                            uint32_t n_valid, float *tau_out, hipStream_t s) {
  BatchSel b{keys, ld, n, nullptr, nullptr, 0, k, tau_out, nullptr, nullptr, nullptr, 0, nullptr, n_valid};
  hipLaunchKernelGGL(batch_select_kernel<false>, dim3(n_queries), dim3(256), 0, s, b);
}

void launch_batch_select_keys(const uint32_t *keys, uint32_t ld, uint32_t n, uint32_t k, uint32_t n_queries,
                              uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_n, uint32_t k_ld, hipStream_t s) {
  BatchSel b{keys, ld, n, nullptr, nullptr, 0, k, nullptr, out_rows, out_keys, out_n, k_ld, nullptr, n_queries};
  hipLaunchKernelGGL(batch_select_kernel<false>, dim3(n_queries), dim3(256), 0, s, b);
}

void launch_batch_select_cand(const void *cand, const uint32_t *cand_count, uint32_t cand_cap, uint32_t k,
                              uint32_t n_queries, uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_n,
                              uint32_t k_ld, uint32_t *overflow, hipStream_t s) {
  BatchSel b{nullptr, 0, 0, (const uint2 *)cand, cand_count, cand_cap, k, nullptr, out_rows, out_keys, out_n, k_ld, overflow, n_queries};
  hipLaunchKernelGGL(batch_select_kernel<true>, dim3(n_queries), dim3(256), 0, s, b);
}

}  // namespace rsgpu
