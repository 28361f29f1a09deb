// scan_kernels.hip -- the FLAT distance scan for gfx950 (MI355X), hand-written HIP.
//
// Replaces the N x dim loop behind VecSimIndex_TopKQuery / VecSimBatchIterator_Next /
// VecSimIndex_RangeQuery (reference call sites src/iterators/hybrid_reader.c:374,417 and
// src/vector_index.c:152) and the per-candidate loop of VecSimIndex_GetDistanceFrom_Unsafe
// (hybrid_reader.c:309-327).
//
// Shape of the work: one pass over a row-contiguous corpus, HBM-bound by ~10x (2-3 flops per 4
// bytes).  Design, MI355X first:
//   * a row is split in 16-byte chunks; a GROUP of G lanes (G = 1..64, power of two) owns a row, lane
//     l reads chunks l, l+G, ...  For dim 768 fp32 a full 64-lane wavefront reads one 3072-byte row as
//     3 x global_load_dwordx4 = 3 x 1 KiB fully coalesced requests;
//   * the query never touches LDS on this path: lane l only ever needs query chunks l, l+G, ..., so
//     they live in registers for the whole kernel (ITERS x 4 VGPRs);
//   * U rows per group are issued back to back before the first use, so a wave keeps U*ITERS 16-byte
//     loads in flight (12 KiB at 768 fp32) and ~7 such waves per SIMD hide the ~2 us HBM latency;
//   * corpus loads are non-temporal: every byte is used exactly once per query;
//   * partial sums are reduced across the group with cross-lane shuffles, the distance becomes an
//     orderable u32 key and lanes 0..U-1 store the tile's keys with one coalesced store.  Keys (4 B
//     per 3072 B read, +0.13 % traffic) feed the radix select in select_kernels.hip, the batch
//     iterator and range queries -- no per-workgroup heaps, no divergent insertion path;
//   * persistent-style grid: blocks_per_cu x 256 CUs blocks of 256 threads, tiles handed out in
//     grid-stride order so that at any moment the whole chip streams one contiguous region.
// MFMA is deliberately NOT used here: a single query is a GEMV (no operand reuse); the batched-query
// GEMM path lives in gemm_kernels.hip.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>

#include "h8_quant.hpp"
#include "kernels.hpp"
#include "scan_ops.hpp"

namespace rsgpu {

ScanTuning &scan_tuning() {
  static ScanTuning t;
  return t;
}

namespace {

// ---- the scan -------------------------------------------------------------------------------------
// TYPE/METRIC: element type and metric.  G lanes per row, ITERS chunks per lane, U rows per group
// per step.  EXACT: chunks == G*ITERS (no chunk masking).  GATHER: rows come from row_ids[] and the
// result is an fp32 distance per candidate instead of a key per row.
template <int TYPE, int METRIC, int G, int ITERS, int U, bool EXACT, bool NT, bool GATHER>
__global__ __launch_bounds__(256) void scan_kernel(const u4 *__restrict__ rows, uint32_t stride16,
                                                   uint32_t chunks, uint32_t row_begin, uint32_t row_end,
                                                   const u4 *__restrict__ query,
                                                   const uint32_t *__restrict__ row_ids,
                                                   typename Tr<TYPE>::key_t *__restrict__ keys,
                                                   typename Tr<TYPE>::out_t *__restrict__ dists) {
  typedef typename Tr<TYPE>::acc_t acc_t;
  typedef typename Tr<TYPE>::out_t out_t;
  constexpr int GPB = 256 / G;  // groups per block
  const uint32_t lane = threadIdx.x % G;
  const uint32_t grp = threadIdx.x / G;

  u4 q[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; i++) {
    uint32_t c = lane + i * G;
    q[i] = (EXACT || c < chunks) ? query[c] : zero4();
  }
  const u4 qx = Tr<TYPE>::kExtra ? query[chunks] : zero4();

  // Row <-> group mapping.  G == 64: a wavefront owns U consecutive rows (one 1 KiB request per
  // chunk, one coalesced U-key store).  G < 64: the block owns GPB*U consecutive rows and row
  // base + u*GPB + grp goes to group grp, so that for a fixed u neighbouring groups read
  // neighbouring rows and every load instruction stays contiguous across the wavefront.
  constexpr bool INTERLEAVE = G < 64;
  // GATHER: the keys slot is unused; a non-NULL pointer there is the DEVICE-side candidate count (the host only knows an
  // upper bound when it enqueues the launch), which then bounds the loop
  if (GATHER && keys) {
    const uint32_t m = *reinterpret_cast<const uint32_t *>(keys);
    if (m < row_end - row_begin) row_end = row_begin + m;
    if (row_end == row_begin) return;
  }
  const uint32_t n = row_end - row_begin;
  const uint32_t rows_per_step = INTERLEAVE ? GPB * U : U;
  const uint32_t n_tiles = (n + rows_per_step - 1) / rows_per_step;
  const uint32_t tile0 = INTERLEAVE ? blockIdx.x : blockIdx.x * GPB + grp;
  const uint32_t tile_step = INTERLEAVE ? gridDim.x : gridDim.x * GPB;
  const uint32_t u_stride = INTERLEAVE ? GPB : 1;

  for (uint32_t tile = tile0; tile < n_tiles; tile += tile_step) {
    const uint32_t r0 = row_begin + tile * rows_per_step + (INTERLEAVE ? grp : 0);
    u4 x[U][ITERS];
    uint32_t rid[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t r = r0 + u * u_stride;
      if (r >= row_end) r = row_end - 1;  // clamp: recomputed, never stored
      bool present = true;
      if (GATHER) {
        r = row_ids[r];
        rid[u] = r;
        present = r != 0xFFFFFFFFu;  // absent label: nothing is read, the result is NaN
        if (!present) r = 0;
      }
      const u4 *p = rows + (size_t)r * stride16;
#pragma unroll
      for (int i = 0; i < ITERS; i++) {
        uint32_t c = lane + i * G;
        x[u][i] = ((EXACT || c < chunks) && (!GATHER || present)) ? load16<NT>(p + c) : zero4();
      }
    }
    out_t d[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      acc_t acc = acc_t();
#pragma unroll
      for (int i = 0; i < ITERS; i++) acc = Op<TYPE, METRIC>::add(acc, x[u][i], q[i]);
      d[u] = finish<TYPE, METRIC>(group_reduce<G>(acc), qx);
    }

    if (INTERLEAVE) {
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          uint32_t r = r0 + u * u_stride;
          if (r < row_end) {
            out_t v = d[u];
            if (GATHER) {
              if (rid[u] == 0xFFFFFFFFu) set_nan(v);
              dists[r] = v;
            } else {
              if (METRIC == KM_IPS || METRIC == KM_L2S) v = (out_t)shadow8_distance<METRIC>((float)v, reinterpret_cast<const float2 *>(row_ids)[r], qx);
              keys[r] = to_key(v);
            }
          }
        }
      }
    } else {
      // lane u of the wavefront stores row r0+u
      out_t mine = d[0];
      uint32_t my_rid = GATHER ? rid[0] : 0;
#pragma unroll
      for (int u = 1; u < U; u++) {
        mine = (lane == (uint32_t)u) ? d[u] : mine;
        if (GATHER) my_rid = (lane == (uint32_t)u) ? rid[u] : my_rid;
      }
      if (lane < (uint32_t)U && r0 + lane < row_end) {
        if (GATHER) {
          if (my_rid == 0xFFFFFFFFu) set_nan(mine);
          dists[r0 + lane] = mine;
        } else {
          if (METRIC == KM_IPS || METRIC == KM_L2S)
            mine = (out_t)shadow8_distance<METRIC>((float)mine, reinterpret_cast<const float2 *>(row_ids)[r0 + lane], qx);
          keys[r0 + lane] = to_key(mine);
        }
      }
    }
  }
}

__device__ __forceinline__ float acc_sum(float a, float b) { return a + b; }
__device__ __forceinline__ double acc_sum(double a, double b) { return a + b; }
__device__ __forceinline__ I2 acc_sum(I2 a, I2 b) { return I2{a.xq + b.xq, a.xx + b.xx}; }

// Fallback for very long rows (more than 512 chunks = 8 KiB): one wavefront per row, runtime loop,
// query re-read through L1/L2 (it is tiny next to the corpus).
template <int TYPE, int METRIC, bool NT, bool GATHER>
__global__ __launch_bounds__(256) void scan_long_kernel(const u4 *__restrict__ rows, uint32_t stride16,
                                                        uint32_t chunks, uint32_t row_begin, uint32_t row_end,
                                                        const u4 *__restrict__ query,
                                                        const uint32_t *__restrict__ row_ids,
                                                        typename Tr<TYPE>::key_t *__restrict__ keys,
                                                        typename Tr<TYPE>::out_t *__restrict__ dists) {
  typedef typename Tr<TYPE>::acc_t acc_t;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t total = gridDim.x * 4;
  const u4 qx = Tr<TYPE>::kExtra ? query[chunks] : zero4();
  if (GATHER && keys) {  // device-side candidate count (see scan_kernel)
    const uint32_t m = *reinterpret_cast<const uint32_t *>(keys);
    if (m < row_end - row_begin) row_end = row_begin + m;
  }
  for (uint32_t t = blockIdx.x * 4 + wave; t < row_end - row_begin; t += total) {
    uint32_t r = row_begin + t, rid = 0;
    if (GATHER) {
      rid = row_ids[r];
      r = rid == 0xFFFFFFFFu ? 0 : rid;
    }
    const u4 *p = rows + (size_t)r * stride16;
    acc_t a0 = acc_t(), a1 = acc_t();
    uint32_t c = lane;
    for (; c + 64 < chunks; c += 128) {
      u4 x0 = load16<NT>(p + c), x1 = load16<NT>(p + c + 64);
      a0 = Op<TYPE, METRIC>::add(a0, x0, query[c]);
      a1 = Op<TYPE, METRIC>::add(a1, x1, query[c + 64]);
    }
    if (c < chunks) a0 = Op<TYPE, METRIC>::add(a0, load16<NT>(p + c), query[c]);
    typename Tr<TYPE>::out_t d = finish<TYPE, METRIC>(group_reduce<64>(acc_sum(a0, a1)), qx);
    if (lane == 0) {
      if (GATHER) {
        if (rid == 0xFFFFFFFFu) set_nan(d);
        dists[row_begin + t] = d;
      } else {
        keys[r] = to_key(d);
      }
    }
  }
}

struct LaunchCtx {
  const u4 *rows;
  uint32_t stride16, chunks, row_begin, row_end;
  const u4 *query;
  const uint32_t *row_ids;
  void *keys;
  void *dists;
  hipStream_t s;
};

// what the last full scan (not gather) launched, for bench.py's roofline.kernel (RSGPU_GetLastScanKernel)
static std::atomic<uint64_t> g_last_scan{0};

template <int TYPE, int METRIC, int G, int ITERS, int U, bool GATHER>
void launch_one(const LaunchCtx &c) {
  const ScanTuning &t = scan_tuning();
  constexpr int GPB = 256 / G;
  uint32_t n = c.row_end - c.row_begin;
  uint32_t need = G < 64 ? (n + GPB * U - 1) / (GPB * U) : ((n + U - 1) / U + GPB - 1) / GPB;
  uint32_t cap = (uint32_t)(t.num_cus * t.blocks_per_cu);
  uint32_t grid = need < cap ? need : cap;
  if (grid == 0) return;
  bool exact = c.chunks == (uint32_t)(G * ITERS);
  // the nontemporal knob is an fp32 A/B experiment; every other type always streams with nt loads
  bool nt = TYPE != KT_F32 || t.nontemporal != 0;
  typename Tr<TYPE>::key_t *keys = (typename Tr<TYPE>::key_t *)c.keys;
  typename Tr<TYPE>::out_t *dists = (typename Tr<TYPE>::out_t *)c.dists;
  if (!GATHER)
    g_last_scan = (uint64_t)TYPE | ((uint64_t)METRIC << 3) | ((uint64_t)G << 6) | ((uint64_t)ITERS << 13) | ((uint64_t)U << 17) |
                  ((uint64_t)exact << 21) | ((uint64_t)nt << 22) | ((uint64_t)grid << 41);
#define RSGPU_LAUNCH(EX, NTV)                                                                              \
  hipLaunchKernelGGL((scan_kernel<TYPE, METRIC, G, ITERS, U, EX, NTV, GATHER>), dim3(grid), dim3(256), 0, c.s, \
                     c.rows, c.stride16, c.chunks, c.row_begin, c.row_end, c.query, c.row_ids, keys, dists)
  if (exact) {
    if (nt) RSGPU_LAUNCH(true, true);
    else if (TYPE == KT_F32) RSGPU_LAUNCH(true, TYPE != KT_F32);
  } else {
    if (nt) RSGPU_LAUNCH(false, true);
    else if (TYPE == KT_F32) RSGPU_LAUNCH(false, TYPE != KT_F32);
  }
#undef RSGPU_LAUNCH
}

template <int TYPE, int METRIC, bool GATHER>
void launch_shape(const LaunchCtx &c) {
  Shape sh = pick_shape(c.chunks);
  int u_over = scan_tuning().rows_per_group;
  if (sh.ITERS > 8) {
    const ScanTuning &t = scan_tuning();
    uint32_t n = c.row_end - c.row_begin;
    uint32_t need = (n + 3) / 4, cap = (uint32_t)(t.num_cus * t.blocks_per_cu);
    uint32_t grid = need < cap ? need : cap;
    if (!grid) return;
    if (!GATHER) g_last_scan = (uint64_t)TYPE | ((uint64_t)METRIC << 3) | (1ull << 40) | ((uint64_t)grid << 41);
    hipLaunchKernelGGL((scan_long_kernel<TYPE, METRIC, true, GATHER>), dim3(grid), dim3(256), 0, c.s, c.rows,
                       c.stride16, c.chunks, c.row_begin, c.row_end, c.query, c.row_ids,
                       (typename Tr<TYPE>::key_t *)c.keys, (typename Tr<TYPE>::out_t *)c.dists);
    return;
  }
  if (sh.ITERS == 1) {
    switch (sh.G) {
      case 1: return launch_one<TYPE, METRIC, 1, 1, 8, GATHER>(c);
      case 2: return launch_one<TYPE, METRIC, 2, 1, 8, GATHER>(c);
      case 4: return launch_one<TYPE, METRIC, 4, 1, 8, GATHER>(c);
      case 8: return launch_one<TYPE, METRIC, 8, 1, 8, GATHER>(c);
      case 16: return launch_one<TYPE, METRIC, 16, 1, 8, GATHER>(c);
      case 32: return launch_one<TYPE, METRIC, 32, 1, 8, GATHER>(c);
      default: return launch_one<TYPE, METRIC, 64, 1, 8, GATHER>(c);
    }
  }
  if (sh.G == 16) {
    switch (sh.ITERS) {
      case 3: return launch_one<TYPE, METRIC, 16, 3, 4, GATHER>(c);
      case 5: return launch_one<TYPE, METRIC, 16, 5, 2, GATHER>(c);
      default: return launch_one<TYPE, METRIC, 16, 7, 2, GATHER>(c);
    }
  }
  if (sh.G == 32) {
    switch (sh.ITERS) {
      case 3: return launch_one<TYPE, METRIC, 32, 3, 4, GATHER>(c);
      case 5: return launch_one<TYPE, METRIC, 32, 5, 2, GATHER>(c);
      default: return launch_one<TYPE, METRIC, 32, 7, 2, GATHER>(c);
    }
  }
  switch (sh.ITERS) {
    case 2: return launch_one<TYPE, METRIC, 64, 2, 4, GATHER>(c);
    case 3:
      // 768 x fp32: U=8 measured 6.67 TB/s vs 6.41 (U=4) / 6.11 (U=2), profiles/r01_tune_scan_*.json
      if (u_over == 2) return launch_one<TYPE, METRIC, 64, 3, 2, GATHER>(c);
      if (u_over == 4) return launch_one<TYPE, METRIC, 64, 3, 4, GATHER>(c);
      return launch_one<TYPE, METRIC, 64, 3, 8, GATHER>(c);
    case 4: return launch_one<TYPE, METRIC, 64, 4, 4, GATHER>(c);
    case 5: return launch_one<TYPE, METRIC, 64, 5, 4, GATHER>(c);
    case 6:
      if (u_over == 2) return launch_one<TYPE, METRIC, 64, 6, 2, GATHER>(c);
      if (u_over == 8) return launch_one<TYPE, METRIC, 64, 6, 8, GATHER>(c);
      return launch_one<TYPE, METRIC, 64, 6, 4, GATHER>(c);
    case 7: return launch_one<TYPE, METRIC, 64, 7, 2, GATHER>(c);
    default: return launch_one<TYPE, METRIC, 64, 8, 2, GATHER>(c);
  }
}

template <bool GATHER>
void dispatch(int type, int metric, const LaunchCtx &c) {
#define RSGPU_CASE(T)                                               \
  case T:                                                           \
    if (metric == KM_L2) launch_shape<T, KM_L2, GATHER>(c);         \
    else launch_shape<T, KM_IP, GATHER>(c);                         \
    break;
#define RSGPU_CASE_INT(T)                                           \
  case T:                                                           \
    if (metric == KM_L2) launch_shape<T, KM_L2, GATHER>(c);         \
    else if (metric == KM_IP) launch_shape<T, KM_IP, GATHER>(c);    \
    else launch_shape<T, KM_COS, GATHER>(c);                        \
    break;
  switch (type) {
    RSGPU_CASE(KT_F32)
    RSGPU_CASE(KT_F64)
    RSGPU_CASE(KT_F16)
    RSGPU_CASE(KT_BF16)
    case KT_I8:
      if (metric == KM_IPS || metric == KM_L2S) {
        if (!GATHER) {
          if (metric == KM_IPS) launch_shape<KT_I8, KM_IPS, false>(c);
          else launch_shape<KT_I8, KM_L2S, false>(c);
        }
        break;
      }
      if (metric == KM_L2) launch_shape<KT_I8, KM_L2, GATHER>(c);
      else if (metric == KM_IP) launch_shape<KT_I8, KM_IP, GATHER>(c);
      else launch_shape<KT_I8, KM_COS, GATHER>(c);
      break;
    RSGPU_CASE_INT(KT_U8)
    default: break;
  }
#undef RSGPU_CASE
#undef RSGPU_CASE_INT
}

// ---- row normalisation (cosine indexes, device bulk loads) -----------------------------------------
template <int TYPE>
__global__ __launch_bounds__(256) void normalize_rows_kernel(u4 *rows, uint32_t stride16, uint32_t chunks,
                                                             uint32_t row_begin, uint32_t row_end) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t total = gridDim.x * 4;
  for (uint32_t r = row_begin + blockIdx.x * 4 + wave; r < row_end; r += total) {
    u4 *p = rows + (size_t)r * stride16;
    typename Tr<TYPE>::acc_t a = 0;
    for (uint32_t c = lane; c < chunks; c += 64) {
      u4 x = p[c];
      a = Op<TYPE, KM_IP>::add(a, x, x);
    }
    a = group_reduce<64>(a);
    if (TYPE == KT_F64) {
      const double n = sqrt((double)a);
      for (uint32_t c = lane; c < chunks; c += 64) {
        u4 x = p[c];
        double v0 = as_d(x.x, x.y) / n, v1 = as_d(x.z, x.w) / n;
        x = (u4){(uint32_t)__double2loint(v0), (uint32_t)__double2hiint(v0), (uint32_t)__double2loint(v1),
                 (uint32_t)__double2hiint(v1)};
        p[c] = x;
      }
      continue;
    }
    float inv = 1.0f / sqrtf((float)a);
    for (uint32_t c = lane; c < chunks; c += 64) {
      u4 x = p[c];
      uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (TYPE == KT_F32) {
          w[j] = __float_as_uint(__uint_as_float(w[j]) * inv);
        } else if (TYPE == KT_F16) {
          half2_t h = as_h2(w[j]);
          h.x = (_Float16)((float)h.x * inv);
          h.y = (_Float16)((float)h.y * inv);
          __builtin_memcpy(&w[j], &h, 4);
        } else {  // bf16, round to nearest even
          uint32_t lo = __float_as_uint(bf_lo(w[j]) * inv), hi = __float_as_uint(bf_hi(w[j]) * inv);
          lo = (lo + 0x7fffu + ((lo >> 16) & 1)) >> 16;
          hi = (hi + 0x7fffu + ((hi >> 16) & 1)) & 0xffff0000u;
          w[j] = hi | lo;
        }
      }
      x = (u4){w[0], w[1], w[2], w[3]};
      p[c] = x;
    }
  }
}

// fp16 (round to nearest even) shadow of fp32 rows, one wavefront per row
__global__ __launch_bounds__(256) void shadow_rows_kernel(const float *__restrict__ rows, uint32_t stride_f,
                                                          uint32_t dim, uint32_t row_begin, uint32_t row_end,
                                                          _Float16 *__restrict__ shadow, uint32_t sstride_h) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t r = row_begin + blockIdx.x * 4 + wave; r < row_end; r += gridDim.x * 4) {
    const float *src = rows + (size_t)r * stride_f;
    _Float16 *dst = shadow + (size_t)r * sstride_h;
    for (uint32_t i = lane; i < sstride_h; i += 64) dst[i] = i < dim ? (_Float16)src[i] : (_Float16)0.0f;
  }
}

// int8 shadow + per-row {scale, |x|^2} of fp32 rows, one wavefront per row
__global__ __launch_bounds__(256) void shadow8_rows_kernel(const float *__restrict__ rows, uint32_t stride_f, uint32_t dim,
                                                           uint32_t row_begin, uint32_t row_end,
                                                           int8_t *__restrict__ shadow, uint32_t sstride,
                                                           float2 *__restrict__ meta, uint32_t *__restrict__ max_bits) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t r = row_begin + blockIdx.x * 4 + wave; r < row_end; r += gridDim.x * 4) {
    const float *src = rows + (size_t)r * stride_f;
    float m = 0.0f, n2 = 0.0f;
    for (uint32_t i = lane; i < dim; i += 64) {
      const float v = src[i];
      m = fmaxf(m, fabsf(v));
      n2 = fmaf(v, v, n2);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      m = fmaxf(m, __shfl_xor(m, o, 64));
      n2 += __shfl_xor(n2, o, 64);
    }
    const float sc = m > 0.0f ? m / 127.0f : 1.0f;  // (an all-zero row: every quantised element is 0)
    int8_t *dst = shadow + (size_t)r * sstride;
    for (uint32_t i = lane; i < sstride; i += 64) {
      float v = i < dim ? rintf(src[i] / sc) : 0.0f;
      v = fminf(fmaxf(v, -127.0f), 127.0f);
      dst[i] = (int8_t)v;
    }
    if (lane == 0) {
      meta[r] = make_float2(sc, n2);
      atomicMax(max_bits, __float_as_uint(sc));
      atomicMax(max_bits + 1, __float_as_uint(n2));
      if (!(n2 < __builtin_inff())) max_bits[2] = 1;  // inf / NaN element (fmaxf would have hidden a NaN)
    }
  }
}

// ---- int8 shadow with ONE index-wide scale, for the batched pass over FLOAT16 indexes (gemm_qs_kernels.hip, KT_I8) ------
// x = s (x8 + ex), q = sq (q8 + eq): the integer dot product s sq (x8 . q8) misses x . q by
//   s sq (x8 . eq + q8 . ex + ex . eq)  <=  s sq (|x8| |eq| + |q8| |ex| + |ex| |eq|)            (Cauchy-Schwarz)
// with the ACTUAL error norms: |eq|, |q8| are the query's own, |x8|, |ex| are bounded by their maxima over the index
// (stats[1], stats[2], atomicMax while the rows are quantised).  One scale for every row keeps the filter an integer
// compare per query.  stats: [0] max |x_i| (f32 bits), [1] max |x8|^2 (u32), [2] max |ex|^2 (f32 bits), [3] non-finite flag.
// bf16 row element (no native type needed: the upper half of an fp32)
struct Bf16 {
  uint16_t bits;
  __device__ __forceinline__ explicit operator float() const { return __uint_as_float((uint32_t)bits << 16); }
};
template <typename T>
struct FiniteMax {  // the largest finite value of the row element type
  static constexpr float v = 65504.0f;
};
template <>
struct FiniteMax<Bf16> {
  static constexpr float v = 3.3895314e38f;  // 0x7F7F
};
template <>
struct FiniteMax<float> {
  static constexpr float v = 3.4028234e38f;
};
template <typename T>
__global__ __launch_bounds__(256) void absmax_rows_kernel(const T *__restrict__ rows, uint32_t stride_h, uint32_t dim,
                                                          uint32_t row_begin, uint32_t row_end, uint32_t *__restrict__ stats) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = 0.0f;
  bool bad = false;
  for (uint32_t r = row_begin + blockIdx.x * 4 + wave; r < row_end; r += gridDim.x * 4) {
    const T *src = rows + (size_t)r * stride_h;
    for (uint32_t i = lane; i < dim; i += 64) {
      const float v = fabsf((float)src[i]);
      bad |= !(v <= FiniteMax<T>::v);
      m = fmaxf(m, v);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) atomicMax(stats, __float_as_uint(m));
  if (bad) stats[3] = 1;
}

// eight consecutive elements of a row (rows are zero padded to whole 16-byte chunks)
__device__ __forceinline__ void load8(const _Float16 *row, uint32_t c, uint32_t stride_e, float out[8]) {
  const u4 x = 8 * c < stride_e ? reinterpret_cast<const u4 *>(row)[c] : zero4();
  const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const half2_t h = as_h2(w[j >> 1]);
    out[j] = (float)((j & 1) ? h.y : h.x);
  }
}
__device__ __forceinline__ void load8(const Bf16 *row, uint32_t c, uint32_t stride_e, float out[8]) {
  const u4 x = 8 * c < stride_e ? reinterpret_cast<const u4 *>(row)[c] : zero4();
  const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    out[2 * j] = __uint_as_float(w[j] << 16);
    out[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ void load8(const float *row, uint32_t c, uint32_t stride_e, float out[8]) {
  const bool in = 8 * c < stride_e;
  const u4 a = in ? reinterpret_cast<const u4 *>(row)[2 * c] : zero4(), b = in ? reinterpret_cast<const u4 *>(row)[2 * c + 1] : zero4();
  out[0] = __uint_as_float(a.x), out[1] = __uint_as_float(a.y), out[2] = __uint_as_float(a.z), out[3] = __uint_as_float(a.w);
  out[4] = __uint_as_float(b.x), out[5] = __uint_as_float(b.y), out[6] = __uint_as_float(b.z), out[7] = __uint_as_float(b.w);
}
template <typename T>
__global__ __launch_bounds__(256) void shadow8g_rows_kernel(const T *__restrict__ rows, uint32_t stride_e, uint32_t dim,
                                                            uint32_t row_begin, uint32_t row_end, float scale,
                                                            int8_t *__restrict__ shadow, uint32_t sstride,
                                                            uint32_t *__restrict__ stats) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv = 1.0f / scale;
  uint32_t n8_max = 0;
  float ne_max = 0.0f;
  // eight elements per lane and step: 16 (32) bytes loaded, 8 bytes stored
  for (uint32_t r = row_begin + blockIdx.x * 4 + wave; r < row_end; r += gridDim.x * 4) {
    const T *src = rows + (size_t)r * stride_e;
    uint2 *dst = reinterpret_cast<uint2 *>(shadow + (size_t)r * sstride);
    uint32_t n8 = 0;
    float ne = 0.0f;
    for (uint32_t c = lane; c < sstride / 8; c += 64) {
      float xf[8];
      load8(src, c, stride_e, xf);
      uint32_t o[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float t = (8 * c + j < dim) ? xf[j] * inv : 0.0f;
        const float v = fminf(fmaxf(rintf(t), -127.0f), 127.0f);
        const float e = t - v;
        const int vi = (int)v;
        o[j >> 2] |= ((uint32_t)vi & 0xffu) << (8 * (j & 3));
        n8 += (uint32_t)(vi * vi);
        ne = fmaf(e, e, ne);
      }
      dst[c] = make_uint2(o[0], o[1]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      n8 += __shfl_xor(n8, o, 64);
      ne += __shfl_xor(ne, o, 64);
    }
    n8_max = n8 > n8_max ? n8 : n8_max;
    ne_max = fmaxf(ne_max, ne);
  }
  if (lane == 0) {
    atomicMax(stats + 1, n8_max);
    atomicMax(stats + 2, __float_as_uint(ne_max));
  }
}

// The error maxima of the IN-FLIGHT quantiser (h8_quant.hpp; gemm_qs_f32_kernel<.., SRC_H8>): the same bookkeeping as
// shadow8g_rows_kernel, nothing stored.  x8 comes from h8_quant2 itself -- the very instruction the pass executes -- and
// ex = x * inv - x8 in fp32 (the product of two fp16 values is exact there).
__global__ __launch_bounds__(256) void h8_stats_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t dim, uint32_t row_begin,
                                                       uint32_t row_end, uint32_t inv2, uint32_t *__restrict__ stats) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv = (float)as_h2(inv2).x;
  uint32_t n8_max = 0;
  float ne_max = 0.0f;
  for (uint32_t r = row_begin + blockIdx.x * 4 + wave; r < row_end; r += gridDim.x * 4) {
    const u4 *src = rows + (size_t)r * stride16;
    uint32_t n8 = 0;
    float ne = 0.0f;
    for (uint32_t c = lane; c < stride16; c += 64) {
      const u4 x = src[c];
      const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t t = h8_quant2(w[j], inv2);
        const half2_t h = as_h2(w[j]);
        const int v0 = (int)(int8_t)(t & 0xffu), v1 = (int)(int8_t)((t >> 16) & 0xffu);
        // (elements past dim are zero padding: x8 = 0, ex = 0)
        const float e0 = (float)h.x * inv - (float)v0, e1 = (float)h.y * inv - (float)v1;
        n8 += (uint32_t)(v0 * v0 + v1 * v1);
        ne = fmaf(e0, e0, ne);
        ne = fmaf(e1, e1, ne);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      n8 += __shfl_xor(n8, o, 64);
      ne += __shfl_xor(ne, o, 64);
    }
    n8_max = n8 > n8_max ? n8 : n8_max;
    ne_max = fmaxf(ne_max, ne);
  }
  if (lane == 0) {
    atomicMax(stats + 1, n8_max);
    atomicMax(stats + 2, __float_as_uint(ne_max));
  }
  (void)dim;
}

// ... and of the fp32 form (f8_quant1): x8 from the very instruction the pass executes, ex = x * inv - x8 (one fma)
__global__ __launch_bounds__(256) void f8_stats_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t dim, uint32_t row_begin,
                                                       uint32_t row_end, float inv, uint32_t *__restrict__ stats) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t n8_max = 0;
  float ne_max = 0.0f;
  for (uint32_t r = row_begin + blockIdx.x * 4 + wave; r < row_end; r += gridDim.x * 4) {
    const u4 *src = rows + (size_t)r * stride16;
    uint32_t n8 = 0;
    float ne = 0.0f;
    for (uint32_t c = lane; c < stride16; c += 64) {
      const u4 x = src[c];
      const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int v = (int)(int8_t)(f8_quant1(w[j], inv) & 0xffu);
        const float e = __builtin_fmaf(__uint_as_float(w[j]), inv, -(float)v);
        n8 += (uint32_t)(v * v);
        ne = fmaf(e, e, ne);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      n8 += __shfl_xor(n8, o, 64);
      ne += __shfl_xor(ne, o, 64);
    }
    n8_max = n8 > n8_max ? n8 : n8_max;
    ne_max = fmaxf(ne_max, ne);
  }
  if (lane == 0) {
    atomicMax(stats + 1, n8_max);
    atomicMax(stats + 2, __float_as_uint(ne_max));
  }
  (void)dim;
}

// The 256 queries of a pass (fp16, already normalised for cosine) -> int8 with their own scales; qscale[q] = s sq
// (distance = 1 - qscale * integer dot) and slack[q] >= 2 E_q, E_q >= |shadow distance - fp32 distance of the fp16 row|:
// the Cauchy-Schwarz band above (x 1.001 for the fp32 arithmetic that computes it) plus the rounding of both distance
// computations, dim 2^-24 |x||q| for the fp32 accumulation of the exact row and a few ulps of the shadow's.
template <typename T>
__global__ __launch_bounds__(64) void quantize_queries_kernel(const T *__restrict__ queries, uint32_t qstride_h, uint32_t dim,
                                                              float scale, const uint32_t *__restrict__ stats,
                                                              int8_t *__restrict__ q8, uint32_t sstride,
                                                              float *__restrict__ qscale, float *__restrict__ slack) {
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  const T *src = queries + (size_t)q * qstride_h;
  float m = 0.0f;
  for (uint32_t i = lane; i < dim; i += 64) m = fmaxf(m, fabsf((float)src[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const bool finite = m <= FiniteMax<T>::v;
  const float sq = (m > 0.0f && finite) ? m / 127.0f : 1.0f, inv = 1.0f / sq;
  uint32_t n8 = 0;
  float ne = 0.0f;
  for (uint32_t i = lane; i < sstride; i += 64) {
    float t = (i < dim && finite) ? (float)src[i] * inv : 0.0f;
    float v = fminf(fmaxf(rintf(t), -127.0f), 127.0f);
    const float e = t - v;
    q8[(size_t)q * sstride + i] = (int8_t)v;
    n8 += (uint32_t)((int)v * (int)v);
    ne = fmaf(e, e, ne);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    n8 += __shfl_xor(n8, o, 64);
    ne += __shfl_xor(ne, o, 64);
  }
  if (lane == 0) {
    const float X8 = sqrtf((float)stats[1]) * 1.0001f, EX = sqrtf(__uint_as_float(stats[2])) * 1.0001f + 1e-3f;
    const float Q8 = sqrtf((float)n8) * 1.0001f, EQ = sqrtf(ne) * 1.0001f + 1e-3f;
    const float ss = scale * sq;
    const float band = ss * (X8 * EQ + Q8 * EX + EX * EQ) * 1.001f;
    const float xq = ss * (X8 + EX) * (Q8 + EQ);  // >= |x| |q| >= |x . q|
    const float round = ((float)dim * 6.0e-8f + 6.0e-7f) * (xq + 1.0f);
    qscale[q] = ss;
    // a query with a non-finite element: an infinite band keeps every row (the exact re-scoring decides)
    // (2 band + 2 round for the shadow-derived bounds; the sample's bound comes from exact rows through the MFMA
    // summation order -- one band and three roundings away -- so four roundings cover both even when the band is 0)
    slack[q] = finite ? 2.0f * band + 4.0f * round : __builtin_inff();
  }
}

// Batched two-stage scan: the candidates (row, shadow key) of every query are re-scored from the fp32 rows with the
// arithmetic of scan_kernel<KT_F32, KM_IP, G, ITERS> (same chunk-to-lane map, same fmaf order, same xor-shuffle
// reduction), so the keys written here are bit-identical to the ones the single-query scan computes for these rows.
// grid = (slices, queries); a group of G lanes per candidate.  Lists that overflowed (count > cap) are skipped: the
// select flags them and the host redoes the query on the single-query path.
template <int TYPE, int METRIC, int G, int ITERS>
__global__ __launch_bounds__(256) void batch_rescore_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t n_rows,
                                                            const u4 *__restrict__ queries, uint32_t qstride16,
                                                            uint2 *__restrict__ cand, const uint32_t *__restrict__ cand_count,
                                                            uint32_t cand_cap, const float *__restrict__ tau, RowBand band, int dense) {
  constexpr int GPB = 256 / G;
  const uint32_t q = blockIdx.y, lane = threadIdx.x % G, grp = threadIdx.x / G;
  const uint32_t cnt = cand_count[q];
  if (cnt > cand_cap) return;
  // candidates collected under the looser bounds of the earlier passes: only those inside the FINAL band
  // (shadow distance <= tau[q]) can be in the answer; the others get the last key and are never re-read
  const uint32_t thr = tau ? f2key(tau[q]) : 0xFFFFFFFFu;
  const float tauf = tau ? tau[q] : __builtin_inff();
  const float hq2 = band.hnorm ? band.hq2[q] : 0.0f;
  u4 qv[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; i++) qv[i] = queries[(size_t)q * qstride16 + lane + i * G];
  uint2 *list = cand + (size_t)q * cand_cap;
  if (dense && !band.hnorm) {
    // A list the last threshold select has pruned (round 6): short, and every entry inside the band -- G consecutive candidates per
    // group would be G re-scorings one after the other in a handful of groups (0.28 ms where the long lists took 0.11).  The
    // groups of the query's slices take the candidates in turn, one row each per step.  Same lanes, same order of operations.
    const uint32_t ng = gridDim.x * GPB;
    for (uint32_t j = blockIdx.x * GPB + grp; j < cnt; j += ng) {
      const uint2 e = list[j];
      const bool inside = e.x < n_rows && e.y <= thr;
      const u4 *p = rows + (size_t)(inside ? e.x : 0u) * stride16;
      u4 x[ITERS];
#pragma unroll
      for (int i = 0; i < ITERS; i++) x[i] = load16<false>(p + lane + i * G);
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < ITERS; i++) acc = Op<TYPE, METRIC>::add(acc, x[i], qv[i]);
      const float d = finish<TYPE, METRIC>(group_reduce<G>(acc), zero4());
      if (lane == 0) list[j].y = inside ? to_key(d) : 0xFFFFFFFFu;
    }
    return;
  }
  // A group looks at G candidates at a time, one per lane (a coalesced read); the few inside the band are then re-scored one
  // after the other by the whole group.  (The lists are long and mostly outside: the int8 band leaves ~10 k collected
  // candidates per query, the first phase of an L2 pass 16 k, of which a few hundred are inside the final band -- walking
  // them one dependent load at a time cost 0.24 ms per batch.)
  for (uint32_t j0 = (blockIdx.x * GPB + grp) * G; j0 < cnt; j0 += gridDim.x * GPB * G) {
    const uint32_t j = j0 + lane;
    uint2 e = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    bool inside = false;
    if (j < cnt) {
      e = list[j];
      if (e.x < n_rows) {  // (row >= n_rows cannot happen; never read outside the corpus)
        if (band.hnorm) {  // the key is an upper bound: the test is on the lower bound it was made from
          // First without the row's norm: |x| <= sqrt(d) + |q| bounds it from the key alone (d <= ub), and most of the
          // list -- collected under the first phases' loose bounds -- is far enough outside to fail even so.
          const float ub = key2f(e.y), hq = hq2 * band.inv2rel;
          const float sx = sqrtf(fmaxf(ub, 0.0f)) + sqrtf(2.0f * hq);
          const float worst = 1.0002f * (band.c1 * (0.5f * sx * sx) + hq2);
          if (!(ub - worst > tauf)) {
            const float lb = ub - 1.0001f * (band.c1 * band.hnorm[e.x] + hq2);
            inside = lb <= tauf || !(lb == lb);
          }
        } else {
          inside = e.y <= thr;
        }
      }
      if (!inside) list[j].y = 0xFFFFFFFFu;
    }
    uint64_t m = __ballot(inside);
    if (G == 32) m = (m >> (32 * (grp & 1))) & 0xFFFFFFFFull;
    while (m) {
      const int src = __builtin_ctzll(m);
      m &= m - 1;
      const uint32_t row = __shfl(e.x, src, G);
      const u4 *p = rows + (size_t)row * stride16;
      u4 x[ITERS];
#pragma unroll
      for (int i = 0; i < ITERS; i++) x[i] = load16<false>(p + lane + i * G);
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < ITERS; i++) acc = Op<TYPE, METRIC>::add(acc, x[i], qv[i]);
      const float d = finish<TYPE, METRIC>(group_reduce<G>(acc), zero4());
      if (lane == 0) list[j0 + src].y = to_key(d);
    }
  }
}

// hn[row] = shrink * |x|^2 / 2 of rows [row_begin, row_end) (fp32; the L2 form of the batched matrix-core pass,
// gemm_qs_kernels.hip); *bad is set if a row's norm is not finite.  One wavefront per row.
template <int TYPE>
__global__ __launch_bounds__(256) void half_norm_rows_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t row_begin,
                                                             uint32_t row_end, float shrink, float *__restrict__ hn,
                                                             uint32_t *__restrict__ bad) {
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t row = row_begin + blockIdx.x * 4 + wv; row < row_end; row += gridDim.x * 4) {
    const u4 *p = rows + (size_t)row * stride16;
    float acc = 0.0f;
    for (uint32_t c = lane; c < stride16; c += 64) {
      const u4 x = load16<true>(p + c);
      acc = Op<TYPE, KM_IP>::add(acc, x, x);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) {
      const float h = 0.5f * acc;
      hn[row] = h * shrink;
      if (!(h <= 3.0e38f)) *bad = 1;  // inf or NaN (plain store: same-address atomics would serialise the kernel)
    }
  }
}

}  // namespace

const char *last_scan_kernel_name(char *buf, size_t cap) {
  static const char *tn[] = {"f32", "f64", "bf16", "f16", "i8", "u8"}, *mn[] = {"L2", "IP", "COS", "IPS", "L2S", "?", "?", "?"};
  const uint64_t v = g_last_scan.load();
  if (!v) {
    snprintf(buf, cap, "none");
    return buf;
  }
  if ((v >> 40) & 1)
    snprintf(buf, cap, "scan_long_kernel<%s,%s,NT=1> grid=%ux256", tn[v & 7], mn[(v >> 3) & 7], (unsigned)(v >> 41));
  else
    snprintf(buf, cap, "scan_kernel<%s,%s,G=%u,ITERS=%u,U=%u,EXACT=%u,NT=%u> grid=%ux256", tn[v & 7], mn[(v >> 3) & 7],
             (unsigned)((v >> 6) & 127), (unsigned)((v >> 13) & 15), (unsigned)((v >> 17) & 15), (unsigned)((v >> 21) & 1),
             (unsigned)((v >> 22) & 1), (unsigned)(v >> 41));
  return buf;
}


bool batch_rescore_supported(uint32_t stride16) {
  const Shape sh = pick_shape(stride16);
  // exact shapes only: the re-scoring kernel has no chunk masking, and a row shape it refuses must be refused HERE --
  // the callers gate the whole shadow route on this answer (a filter pass whose survivors are never re-scored would
  // hand shadow distances back as exact ones)
  if ((uint32_t)(sh.G * sh.ITERS) != stride16) return false;
  return (sh.G == 64 && sh.ITERS >= 1 && sh.ITERS <= 4) || (sh.G == 32 && (sh.ITERS == 1 || sh.ITERS == 3));
}

bool launch_batch_rescore(const void *rows, size_t stride, uint32_t n_rows, const void *queries, size_t qstride, void *cand,
                          const uint32_t *cand_count, uint32_t cand_cap, uint32_t n_queries, const float *tau, hipStream_t s,
                          int type, int metric, const RowBand *band, bool dense) {
  const uint32_t s16 = (uint32_t)(stride / 16);
  if (!batch_rescore_supported(s16) || !n_queries || (type != KT_F32 && type != KT_F16 && type != KT_BF16)) return false;
  if (metric != KM_IP && metric != KM_L2) return false;
  const RowBand rb = band ? *band : RowBand{};
  const Shape sh = pick_shape(s16);
  if ((uint32_t)(sh.G * sh.ITERS) != s16) return false;  // exact shapes only (no chunk masking here)
  // (64 slices per query: a slice walks its share of the list with one dependent load per candidate, most of which it
  // skips -- the int8 band leaves ~10 k collected candidates per query of which a few hundred are inside the final band)
  const dim3 grid(64, n_queries), block(256);
#define RSGPU_RESCORE_M(TT, MM, GG, II)                                                                                    \
  hipLaunchKernelGGL((batch_rescore_kernel<TT, MM, GG, II>), grid, block, 0, s, (const u4 *)rows, s16, n_rows,             \
                     (const u4 *)queries, (uint32_t)(qstride / 16), (uint2 *)cand, cand_count, cand_cap, tau, rb, dense ? 1 : 0)
#define RSGPU_RESCORE_T(TT, GG, II)                                                                                        \
  do {                                                                                                                     \
    if (metric == KM_L2) RSGPU_RESCORE_M(TT, KM_L2, GG, II);                                                               \
    else RSGPU_RESCORE_M(TT, KM_IP, GG, II);                                                                               \
  } while (0)
#define RSGPU_RESCORE(GG, II)                                                                                              \
  do {                                                                                                                     \
    if (type == KT_F16) RSGPU_RESCORE_T(KT_F16, GG, II);                                                                   \
    else if (type == KT_BF16) RSGPU_RESCORE_T(KT_BF16, GG, II);                                                            \
    else RSGPU_RESCORE_T(KT_F32, GG, II);                                                                                  \
  } while (0)
  if (sh.G == 32) {
    if (sh.ITERS == 1) RSGPU_RESCORE(32, 1);
    else RSGPU_RESCORE(32, 3);
  } else {
    switch (sh.ITERS) {
      case 1: RSGPU_RESCORE(64, 1); break;
      case 2: RSGPU_RESCORE(64, 2); break;
      case 3: RSGPU_RESCORE(64, 3); break;
      default: RSGPU_RESCORE(64, 4); break;
    }
  }
#undef RSGPU_RESCORE
#undef RSGPU_RESCORE_T
#undef RSGPU_RESCORE_M
  return true;
}

void launch_half_norm_rows(int type, const void *rows, size_t stride, uint32_t row_begin, uint32_t row_end, float shrink,
                           float *hn, uint32_t *bad, hipStream_t s) {
  if (row_end <= row_begin) return;
  const uint32_t n = row_end - row_begin, need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 16);
  const dim3 grid(need < cap ? need : cap), block(256);
  const uint32_t s16 = (uint32_t)(stride / 16);
  if (type == KT_F32) hipLaunchKernelGGL(half_norm_rows_kernel<KT_F32>, grid, block, 0, s, (const u4 *)rows, s16, row_begin, row_end, shrink, hn, bad);
  else if (type == KT_BF16) hipLaunchKernelGGL(half_norm_rows_kernel<KT_BF16>, grid, block, 0, s, (const u4 *)rows, s16, row_begin, row_end, shrink, hn, bad);
  else hipLaunchKernelGGL(half_norm_rows_kernel<KT_F16>, grid, block, 0, s, (const u4 *)rows, s16, row_begin, row_end, shrink, hn, bad);
}

void launch_absmax_rows(int type, const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end,
                        uint32_t *stats, hipStream_t s) {
  if (row_end <= row_begin) return;
  const uint32_t n = row_end - row_begin, need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 8);
  const dim3 grid(need < cap ? need : cap), block(256);
  if (type == KT_F32)
    hipLaunchKernelGGL(absmax_rows_kernel<float>, grid, block, 0, s, (const float *)rows, (uint32_t)(stride / 4), dim, row_begin,
                       row_end, stats);
  else if (type == KT_BF16)
    hipLaunchKernelGGL(absmax_rows_kernel<Bf16>, grid, block, 0, s, (const Bf16 *)rows, (uint32_t)(stride / 2), dim, row_begin,
                       row_end, stats);
  else
    hipLaunchKernelGGL(absmax_rows_kernel<_Float16>, grid, block, 0, s, (const _Float16 *)rows, (uint32_t)(stride / 2), dim,
                       row_begin, row_end, stats);
}
void launch_shadow8g_rows(int type, const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end,
                          float scale, void *shadow, size_t sstride, uint32_t *stats, hipStream_t s) {
  if (row_end <= row_begin) return;
  const uint32_t n = row_end - row_begin, need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 8);
  const dim3 grid(need < cap ? need : cap), block(256);
  if (type == KT_F32)
    hipLaunchKernelGGL(shadow8g_rows_kernel<float>, grid, block, 0, s, (const float *)rows, (uint32_t)(stride / 4), dim, row_begin,
                       row_end, scale, (int8_t *)shadow, (uint32_t)sstride, stats);
  else if (type == KT_BF16)
    hipLaunchKernelGGL(shadow8g_rows_kernel<Bf16>, grid, block, 0, s, (const Bf16 *)rows, (uint32_t)(stride / 2), dim, row_begin,
                       row_end, scale, (int8_t *)shadow, (uint32_t)sstride, stats);
  else
    hipLaunchKernelGGL(shadow8g_rows_kernel<_Float16>, grid, block, 0, s, (const _Float16 *)rows, (uint32_t)(stride / 2), dim,
                       row_begin, row_end, scale, (int8_t *)shadow, (uint32_t)sstride, stats);
}
void launch_f8_stats(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, float inv, uint32_t *stats,
                     hipStream_t s) {
  if (row_end <= row_begin) return;
  const uint32_t n = row_end - row_begin, need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 8);
  hipLaunchKernelGGL(f8_stats_kernel, dim3(need < cap ? need : cap), dim3(256), 0, s, (const u4 *)rows, (uint32_t)(stride / 16), dim, row_begin,
                     row_end, inv, stats);
}
void launch_h8_stats(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, uint16_t inv_h_bits,
                     uint32_t *stats, hipStream_t s) {
  if (row_end <= row_begin) return;
  const uint32_t n = row_end - row_begin, need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 8);
  hipLaunchKernelGGL(h8_stats_kernel, dim3(need < cap ? need : cap), dim3(256), 0, s, (const u4 *)rows, (uint32_t)(stride / 16), dim,
                     row_begin, row_end, (uint32_t)inv_h_bits | ((uint32_t)inv_h_bits << 16), stats);
}
void launch_quantize_queries(int type, const void *queries, size_t qstride, uint32_t dim, uint32_t n_queries, float scale,
                             const uint32_t *stats, void *q8, size_t sstride, float *qscale, float *slack, hipStream_t s) {
  if (!n_queries) return;
  if (type == KT_F32)
    hipLaunchKernelGGL(quantize_queries_kernel<float>, dim3(n_queries), dim3(64), 0, s, (const float *)queries,
                       (uint32_t)(qstride / 4), dim, scale, stats, (int8_t *)q8, (uint32_t)sstride, qscale, slack);
  else if (type == KT_BF16)
    hipLaunchKernelGGL(quantize_queries_kernel<Bf16>, dim3(n_queries), dim3(64), 0, s, (const Bf16 *)queries,
                       (uint32_t)(qstride / 2), dim, scale, stats, (int8_t *)q8, (uint32_t)sstride, qscale, slack);
  else
    hipLaunchKernelGGL(quantize_queries_kernel<_Float16>, dim3(n_queries), dim3(64), 0, s, (const _Float16 *)queries,
                       (uint32_t)(qstride / 2), dim, scale, stats, (int8_t *)q8, (uint32_t)sstride, qscale, slack);
}

void launch_shadow8_rows(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, void *shadow,
                         size_t sstride, float *meta, uint32_t *max_bits, hipStream_t s) {
  if (row_end <= row_begin) return;
  uint32_t n = row_end - row_begin;
  uint32_t need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 8);
  hipLaunchKernelGGL(shadow8_rows_kernel, dim3(need < cap ? need : cap), dim3(256), 0, s, (const float *)rows,
                     (uint32_t)(stride / 4), dim, row_begin, row_end, (int8_t *)shadow, (uint32_t)sstride, (float2 *)meta, max_bits);
}

void launch_shadow_rows(const void *rows, size_t stride, uint32_t dim, uint32_t row_begin, uint32_t row_end, void *shadow,
                        size_t sstride, hipStream_t s) {
  if (row_end <= row_begin) return;
  uint32_t n = row_end - row_begin;
  uint32_t need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 8);
  hipLaunchKernelGGL(shadow_rows_kernel, dim3(need < cap ? need : cap), dim3(256), 0, s, (const float *)rows,
                     (uint32_t)(stride / 4), dim, row_begin, row_end, (_Float16 *)shadow, (uint32_t)(sstride / 2));
}

void launch_scan(const void *rows, size_t stride, uint32_t dim, int type, int metric, uint32_t row_begin,
                 uint32_t row_end, const void *query, void *keys, hipStream_t s, const float *row_meta) {
  (void)dim;
  if (row_end <= row_begin) return;
  // (KM_IPS / KM_L2S: the per-row {scale, |x|^2} travel in the row_ids slot, which a plain scan does not use)
  LaunchCtx c{(const u4 *)rows, (uint32_t)(stride / 16), (uint32_t)(stride / 16), row_begin, row_end,
              (const u4 *)query, reinterpret_cast<const uint32_t *>(row_meta), keys, nullptr, s};
  dispatch<false>(type, metric, c);
}

void launch_gather(const void *rows, size_t stride, uint32_t dim, int type, int metric, const uint32_t *row_ids,
                   uint32_t m, const void *query, void *out, hipStream_t s, const uint32_t *m_dev) {
  (void)dim;
  if (!m) return;
  LaunchCtx c{(const u4 *)rows, (uint32_t)(stride / 16), (uint32_t)(stride / 16), 0, m,
              (const u4 *)query, row_ids, const_cast<uint32_t *>(m_dev), out, s};
  dispatch<true>(type, metric, c);
}

void launch_normalize_rows(void *rows, size_t stride, uint32_t dim, int type, uint32_t row_begin, uint32_t row_end,
                           hipStream_t s) {
  (void)dim;
  if (row_end <= row_begin) return;
  uint32_t n = row_end - row_begin;
  uint32_t need = (n + 3) / 4, cap = (uint32_t)(scan_tuning().num_cus * 8);
  uint32_t grid = need < cap ? need : cap;
  uint32_t s16 = (uint32_t)(stride / 16);
  switch (type) {
    case KT_F32:
      hipLaunchKernelGGL(normalize_rows_kernel<KT_F32>, dim3(grid), dim3(256), 0, s, (u4 *)rows, s16, s16, row_begin, row_end);
      break;
    case KT_F16:
      hipLaunchKernelGGL(normalize_rows_kernel<KT_F16>, dim3(grid), dim3(256), 0, s, (u4 *)rows, s16, s16, row_begin, row_end);
      break;
    case KT_BF16:
      hipLaunchKernelGGL(normalize_rows_kernel<KT_BF16>, dim3(grid), dim3(256), 0, s, (u4 *)rows, s16, s16, row_begin, row_end);
      break;
    case KT_F64:
      hipLaunchKernelGGL(normalize_rows_kernel<KT_F64>, dim3(grid), dim3(256), 0, s, (u4 *)rows, s16, s16, row_begin, row_end);
      break;
    default: break;  // INT8/UINT8 rows stay as they are (KM_COS divides by the norms)
  }
}

}  // namespace rsgpu
