// scan_mq_kernels.hip -- the FLAT distance scan for SEVERAL queries per corpus pass (gfx950, hand-written HIP).
//
// Why it exists.  VecSim's C ABI answers one query per call (reference src/iterators/hybrid_reader.c:374) and RediSearch
// issues those calls from N worker threads (src/util/workers.c:58,104).  N concurrent single-query scans are N
// independent passes over the same 30 GB sharing one HBM: the aggregate stays at one pass per ~4.8 ms however many
// callers there are.  The coalescer in flat_index.cpp lets queries that arrive while a pass is in flight join the next
// pass; this kernel is that pass: every row is read ONCE and scored against up to 8 queries held in registers (up to 16
// FLOAT32 queries held in LDS: scan_mq16_kernel).
//
// Bit-identity with scan_kernel is by construction, not by tolerance (scan_ops.hpp):
//   * same chunk-to-lane map (lane l of a G-lane group owns chunks l, l+G, ...), same per-lane operation order
//     (chunk by chunk, element by element, one fused multiply-add per element);
//   * same reduction tree: scan_kernel folds a group with v += shfl_xor(v, m) for m = G/2 ... 1.  Here a lane holds
//     V = U*B partial sums (U rows x B queries); at mask m a lane pairs value i with value i + V/2, KEEPS the one its
//     own bit of m selects and hands the other to its partner, so the number of live values halves per step while every
//     surviving value is still "own partial + partner's partial" for exactly the lane pairs of the butterfly above --
//     fp32 addition is commutative, so the bits are the butterfly's.  V/2 + V/4 + ... ~ V exchanges for V sums instead
//     of 6 V; masks 32 and 16 are one v_permlane32_swap / v_permlane16_swap (gfx950) per PAIR of values.
//   * FLOAT32: two queries share one v_pk_fma_f32 (the row element broadcast through op_sel, the two queries' elements
//     in a register pair) -- each query's accumulator still sees its products in the single-query order.
// HBM-bound like the single-query scan: B = 8 costs ~70 VALU instructions per row and wavefront next to 3 KiB of loads.
// Measured at 10 M x 768 fp32 (profiles/r03_mq_scan_ab.txt): 4.68-4.79 ms per pass with 2-4 queries, 5.07 ms with 8
// (6.06 TB/s = 76 % of the HBM peak) against 4.71-4.78 ms for the single-query scan.  Loads must stay UNCONDITIONAL (see
// load_chunk): the first version predicated them and hipcc serialised every one behind its own s_waitcnt vmcnt(0) --
// 8.7 ms.  A variant that streamed the rows through per-wave LDS rings by DMA (global_load_lds, no destination registers,
// five slots in flight per wave) was built, was bit-identical, and lost: 5.26-5.38 ms with 8 queries, 4.80-5.00 with 4 --
// removed.  Keys go to keys[b * keys_ld + row], one array per query, and feed the same selection kernels.
// Nine to sixteen FLOAT32 queries: scan_mq16_kernel below (queries in LDS): 5.85-5.94 ms per sixteen-query pass against
// 6.48-6.53 with the queries in registers (scan_mq_kernel<..., U = 2, B = 16>, kept as knob mq16 = 2) and 10.4 for two
// passes of eight; fp16 / bf16: two launches of up to eight.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>

#include "kernels.hpp"
#include "scan_ops.hpp"

namespace rsgpu {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }
constexpr int kMq16U3 = 2;  // rows per step of the sixteen-query kernel at three chunks per lane (dim 768 fp32)

// chunk c of a row / query.  EXACT: every lane's chunks exist (chunks == G * ITERS).  Otherwise the load is still
// UNCONDITIONAL -- from the last chunk -- and the value is dropped afterwards: a predicated load becomes a branch with
// its own s_waitcnt vmcnt(0), i.e. one load in flight at a time.
template <bool EXACT, bool NT>
__device__ __forceinline__ u4 load_chunk(const u4 *__restrict__ p, uint32_t c, uint32_t chunks) {
  if (EXACT) return load16<NT>(p + c);
  const u4 t = load16<NT>(p + (c < chunks ? c : chunks - 1));
  return c < chunks ? t : zero4();
}

// ---- the halving butterfly over a group of G lanes: M = current mask, CUR = live values per lane ----------------------
__device__ __forceinline__ uint32_t as_bits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ uint32_t as_bits(int i) { return (uint32_t)i; }
template <typename T>
__device__ __forceinline__ T from_bits(uint32_t u);
template <>
__device__ __forceinline__ float from_bits<float>(uint32_t u) { return __uint_as_float(u); }
template <>
__device__ __forceinline__ int from_bits<int>(uint32_t u) { return (int)u; }

template <int M, int CUR, typename T = float>
struct MqRed {
  static __device__ __forceinline__ void run(T *v, uint32_t gl) {
    if constexpr (CUR >= 2) {
      constexpr int H = CUR / 2;
#pragma unroll
      for (int i = 0; i < H; i++) {
        const T a = v[i], b = v[i + H];
        if constexpr (M == 32 && sizeof(T) == 4) {
          // a' = {a.lo32, b.lo32}, b' = {a.hi32, b.hi32}: lanes < 32 get a[l] + a[l+32], lanes >= 32 b[l-32] + b[l]
          const u2v r = __builtin_amdgcn_permlane32_swap(as_bits(a), as_bits(b), false, false);
          v[i] = from_bits<T>(r.x) + from_bits<T>(r.y);
        } else if constexpr (M == 16 && sizeof(T) == 4) {
          // odd 16-lane rows of a <-> even rows of b: even rows get a[l] + a[l+16], odd rows b[l-16] + b[l]
          const u2v r = __builtin_amdgcn_permlane16_swap(as_bits(a), as_bits(b), false, false);
          v[i] = from_bits<T>(r.x) + from_bits<T>(r.y);
        } else {
          const bool hi = (gl & (uint32_t)M) != 0;
          const T keep = hi ? b : a, send = hi ? a : b;
          v[i] = keep + __shfl_xor(send, M, 64);
        }
      }
      if constexpr (M > 1) MqRed<M / 2, H, T>::run(v, gl);
    } else {
      v[0] += __shfl_xor(v[0], M, 64);
      if constexpr (M > 1) MqRed<M / 2, 1, T>::run(v, gl);
    }
  }
};

// ---- B queries in registers + the per-lane partial sums of U rows against them -----------------------------------------
// generic: one Op<TYPE, METRIC>::add per (row, query, chunk) -- fp16 / bf16
template <int TYPE, int METRIC, int ITERS, int B>
struct MqQ {
  u4 q[B][ITERS];
  template <bool EXACT>
  __device__ __forceinline__ void load(const u4 *__restrict__ queries, uint32_t qstride16, uint32_t lane, uint32_t G,
                                       uint32_t chunks, uint32_t nq) {
#pragma unroll
    for (int b = 0; b < B; b++) {
      const uint32_t bb = (uint32_t)b < nq ? (uint32_t)b : nq - 1;  // unused slots repeat the last query, never stored
#pragma unroll
      for (int i = 0; i < ITERS; i++) {
        const uint32_t c = lane + i * G;
        q[b][i] = load_chunk<EXACT, false>(queries + (size_t)bb * qstride16, c, chunks);
      }
    }
  }
  // the B per-lane partial sums of one row
  __device__ __forceinline__ void partial_row(const u4 (&x)[ITERS], float (&out)[B]) {
#pragma unroll
    for (int b = 0; b < B; b++) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < ITERS; i++) {
        // (opaque: otherwise the widening of the query's halves -- loop invariant -- is hoisted out of the row loop and
        // the queries take twice the registers, as floats)
        asm("" : "+v"(q[b][i]));
        acc = Op<TYPE, METRIC>::add(acc, x[i], q[b][i]);
      }
      out[b] = acc;
    }
  }
};
// FLOAT32: query pairs in register pairs, one packed FMA per element and pair
template <int METRIC, int ITERS, int B>
struct MqQ<KT_F32, METRIC, ITERS, B> {
  f2 qp[ITERS][B / 2][4];
  template <bool EXACT>
  __device__ __forceinline__ void load(const u4 *__restrict__ queries, uint32_t qstride16, uint32_t lane, uint32_t G,
                                       uint32_t chunks, uint32_t nq) {
#pragma unroll
    for (int b = 0; b < B; b++) {
      const uint32_t bb = (uint32_t)b < nq ? (uint32_t)b : nq - 1;
#pragma unroll
      for (int i = 0; i < ITERS; i++) {
        const uint32_t c = lane + i * G;
        const u4 t = load_chunk<EXACT, false>(queries + (size_t)bb * qstride16, c, chunks);
        const float e[4] = {__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (b & 1) qp[i][b / 2][j].y = e[j];
          else qp[i][b / 2][j].x = e[j];
        }
      }
    }
  }
  __device__ __forceinline__ void partial_row(const u4 (&x)[ITERS], float (&out)[B]) {
    f2 acc[B / 2];
#pragma unroll
    for (int p = 0; p < B / 2; p++) acc[p] = (f2){0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
      const float e[4] = {__uint_as_float(x[i].x), __uint_as_float(x[i].y), __uint_as_float(x[i].z), __uint_as_float(x[i].w)};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f2 xs = (f2){e[j], e[j]};
#pragma unroll
        for (int p = 0; p < B / 2; p++) {
          if (METRIC == KM_L2) {
            const f2 d = xs - qp[i][p][j];
            acc[p] = __builtin_elementwise_fma(d, d, acc[p]);
          } else {
            acc[p] = __builtin_elementwise_fma(xs, qp[i][p][j], acc[p]);
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < B / 2; p++) {
      out[2 * p] = acc[p].x;
      out[2 * p + 1] = acc[p].y;
    }
  }
};

// TYPE / METRIC / G / ITERS as in scan_kernel (scan_ops.hpp pick_shape); U rows per group and step; B query slots.
template <int TYPE, int METRIC, int G, int ITERS, int U, int B, bool EXACT>
__global__ __launch_bounds__(256, 2) void scan_mq_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t chunks,
                                                         uint32_t row_begin, uint32_t row_end,
                                                         const u4 *__restrict__ queries, uint32_t qstride16, uint32_t nq,
                                                         uint32_t *__restrict__ keys, uint32_t keys_ld) {
  constexpr int GPB = 256 / G;
  constexpr int V = U * B, LG = ilog2(G), LV = ilog2(V);
  constexpr int H = LV < LG ? LV : LG;   // halving steps
  constexpr int CNT = V >> H;            // finished sums per lane
  const uint32_t lane = threadIdx.x % G;
  const uint32_t grp = threadIdx.x / G;

  MqQ<TYPE, METRIC, ITERS, B> q;
  q.template load<EXACT>(queries, qstride16, lane, G, chunks, nq);

  constexpr bool INTERLEAVE = G < 64;  // (row <-> group mapping of scan_kernel)
  const uint32_t n = row_end - row_begin;
  const uint32_t rows_per_step = INTERLEAVE ? GPB * U : U;
  const uint32_t n_tiles = (n + rows_per_step - 1) / rows_per_step;
  const uint32_t tile0 = INTERLEAVE ? blockIdx.x : blockIdx.x * GPB + grp;
  const uint32_t tile_step = INTERLEAVE ? gridDim.x : gridDim.x * GPB;
  const uint32_t u_stride = INTERLEAVE ? GPB : 1;
  // which finished sums this lane ends up with: the top H bits of its group lane are the top H bits of j = b * U + u
  const uint32_t jtop = lane >> (LG - H);
  const bool writer = (lane & ((1u << (LG - H)) - 1u)) == 0;

  for (uint32_t tile = tile0; tile < n_tiles; tile += tile_step) {
    const uint32_t r0 = row_begin + tile * rows_per_step + (INTERLEAVE ? grp : 0);
    u4 x[U][ITERS];
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t r = r0 + u * u_stride;
      if (r >= row_end) r = row_end - 1;  // clamp: recomputed, never stored
      const u4 *p = rows + (size_t)r * stride16;
#pragma unroll
      for (int i = 0; i < ITERS; i++) {
        const uint32_t c = lane + i * G;
        x[u][i] = load_chunk<EXACT, true>(p, c, chunks);
      }
    }
    float v[V];
#pragma unroll
    for (int u = 0; u < U; u++) {
      float pr[B];
      q.partial_row(x[u], pr);
#pragma unroll
      for (int b = 0; b < B; b++) v[b * U + u] = pr[b];
    }
    if constexpr (G > 1) MqRed<G / 2, V>::run(v, lane);
#pragma unroll
    for (int p = 0; p < CNT; p++) {
      const uint32_t j = jtop * CNT + p, b = j / U, u = j % U;
      const uint32_t r = r0 + u * u_stride;
      if (writer && b < nq && r < row_end) keys[(size_t)b * keys_ld + r] = to_key(finish<TYPE, METRIC>(v[p], zero4()));
    }
  }
}

// SIXTEEN FLOAT32 queries per pass: 192 query registers per lane at dim 768 leave room for two rows in flight per wave,
// and a pass became latency-bound (6.5 ms, 4.7 TB/s).  Here the queries live in LDS, once per workgroup (every G-lane
// group needs the same chunks), in the register-pair layout of the packed FMA -- 48 KiB at dim 768 -- and a lane fetches
// a pair's chunk with two ds_read_b128 right where it is multiplied, for all U rows of the step at once: the registers
// hold rows and sums only.  Per (row, query) the products still arrive chunk by chunk, element by element, and the
// reduction is the same halving butterfly: bit for bit scan_kernel's keys.
template <int METRIC, int G, int ITERS, int U, bool EXACT>
__global__ __launch_bounds__(256, 2) void scan_mq16_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t chunks,
                                                           uint32_t row_begin, uint32_t row_end,
                                                           const u4 *__restrict__ queries, uint32_t qstride16, uint32_t nq,
                                                           uint32_t *__restrict__ keys, uint32_t keys_ld) {
  constexpr int B = 16, P = B / 2, GPB = 256 / G;
  constexpr int V = U * B, LG = ilog2(G), LV = ilog2(V);
  constexpr int H = LV < LG ? LV : LG;
  constexpr int CNT = V >> H;
  __shared__ u4 qs[ITERS * P * 2 * G];  // [(i * P + p) * 2 + h][lane] = {q2p[4i'+2h], q2p+1[..], q2p[4i'+2h+1], q2p+1[..]}
  const uint32_t lane = threadIdx.x % G;
  const uint32_t grp = threadIdx.x / G;
  for (uint32_t idx = threadIdx.x; idx < (uint32_t)(ITERS * P * G); idx += 256) {
    const uint32_t l = idx % G, ip = idx / G, i = ip / P, p = ip % P;
    const uint32_t b0 = 2 * p < nq ? 2 * p : nq - 1, b1 = 2 * p + 1 < nq ? 2 * p + 1 : nq - 1;  // unused slots repeat the last query
    const uint32_t c = l + i * G;
    const u4 t0 = load_chunk<EXACT, false>(queries + (size_t)b0 * qstride16, c, chunks);
    const u4 t1 = load_chunk<EXACT, false>(queries + (size_t)b1 * qstride16, c, chunks);
    qs[(ip * 2 + 0) * G + l] = (u4){t0.x, t1.x, t0.y, t1.y};
    qs[(ip * 2 + 1) * G + l] = (u4){t0.z, t1.z, t0.w, t1.w};
  }
  __syncthreads();

  constexpr bool INTERLEAVE = G < 64;  // (row <-> group mapping of scan_kernel)
  const uint32_t n = row_end - row_begin;
  const uint32_t rows_per_step = INTERLEAVE ? GPB * U : U;
  const uint32_t n_tiles = (n + rows_per_step - 1) / rows_per_step;
  const uint32_t tile0 = INTERLEAVE ? blockIdx.x : blockIdx.x * GPB + grp;
  const uint32_t tile_step = INTERLEAVE ? gridDim.x : gridDim.x * GPB;
  const uint32_t u_stride = INTERLEAVE ? GPB : 1;
  const uint32_t jtop = lane >> (LG - H);
  const bool writer = (lane & ((1u << (LG - H)) - 1u)) == 0;

  for (uint32_t tile = tile0; tile < n_tiles; tile += tile_step) {
    const uint32_t r0 = row_begin + tile * rows_per_step + (INTERLEAVE ? grp : 0);
    u4 x[U][ITERS];
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t r = r0 + u * u_stride;
      if (r >= row_end) r = row_end - 1;  // clamp: recomputed, never stored
      const u4 *p = rows + (size_t)r * stride16;
#pragma unroll
      for (int i = 0; i < ITERS; i++) x[u][i] = load_chunk<EXACT, true>(p, lane + i * G, chunks);
    }
    f2 acc[U][P];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int p = 0; p < P; p++) acc[u][p] = (f2){0.0f, 0.0f};
    // (opaque per tile: the query reads do not depend on the tile, and hipcc would hoist all of them out of the row loop --
    // 192 registers, the thing this kernel exists to avoid)
    uint32_t ql = lane;
    asm volatile("" : "+v"(ql));
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
      for (int p = 0; p < P; p++) {
        const u4 qa = qs[((i * P + p) * 2 + 0) * G + ql], qb = qs[((i * P + p) * 2 + 1) * G + ql];
        const f2 qj[4] = {(f2){__uint_as_float(qa.x), __uint_as_float(qa.y)}, (f2){__uint_as_float(qa.z), __uint_as_float(qa.w)},
                          (f2){__uint_as_float(qb.x), __uint_as_float(qb.y)}, (f2){__uint_as_float(qb.z), __uint_as_float(qb.w)}};
#pragma unroll
        for (int u = 0; u < U; u++) {
          const float e[4] = {__uint_as_float(x[u][i].x), __uint_as_float(x[u][i].y), __uint_as_float(x[u][i].z),
                              __uint_as_float(x[u][i].w)};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const f2 xs = (f2){e[j], e[j]};
            if (METRIC == KM_L2) {
              const f2 d = xs - qj[j];
              acc[u][p] = __builtin_elementwise_fma(d, d, acc[u][p]);
            } else {
              acc[u][p] = __builtin_elementwise_fma(xs, qj[j], acc[u][p]);
            }
          }
        }
      }
    }
    float v[V];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int p = 0; p < P; p++) {
        v[(2 * p) * U + u] = acc[u][p].x;
        v[(2 * p + 1) * U + u] = acc[u][p].y;
      }
    if constexpr (G > 1) MqRed<G / 2, V>::run(v, lane);
#pragma unroll
    for (int p = 0; p < CNT; p++) {
      const uint32_t j = jtop * CNT + p, b = j / U, u = j % U;
      const uint32_t r = r0 + u * u_stride;
      if (writer && b < nq && r < row_end) keys[(size_t)b * keys_ld + r] = to_key(finish<KT_F32, METRIC>(v[p], zero4()));
    }
  }
}

// The int8 shadow (FLOAT32 indexes created with shadow8: int8 rows with {scale, |x|^2} per row, flat_index.hpp) scanned for
// up to EIGHT queries per pass: the filter pass of the two-stage exact scan for coalesced callers.  Rows are dim bytes
// (16-byte chunks, 16 lanes per row, ITERS = chunks / 16), the queries' int8 copies sit in registers, v_dot4_i32_i8 sums
// are exact integers -- whatever the order -- and the key is the single scan's shadow8_distance of the same integer, so the
// keys ARE scan_kernel<KT_I8, KM_IPS / KM_L2S>'s.  qx[b] = {query scale, |q|^2}.
template <int METRIC, int ITERS, int U, int B>
__global__ __launch_bounds__(256, 2) void scan_mq_i8_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t row_begin,
                                                            uint32_t row_end, const float2 *__restrict__ row_meta,
                                                            const u4 *__restrict__ queries, uint32_t qstride16,
                                                            const float2 *__restrict__ qx, uint32_t nq,
                                                            uint32_t *__restrict__ keys, uint32_t keys_ld) {
  constexpr int G = 16, GPB = 256 / G;
  constexpr int V = U * B, LG = 4, LV = ilog2(V);
  constexpr int H = LV < LG ? LV : LG;
  constexpr int CNT = V >> H;
  static_assert(LV >= LG, "every lane of a group finishes at least one sum");
  __shared__ float2 qxs[B];
  const uint32_t lane = threadIdx.x % G, grp = threadIdx.x / G;
  if (threadIdx.x < B) qxs[threadIdx.x] = qx[threadIdx.x < nq ? threadIdx.x : nq - 1];
  u4 q[B][ITERS];
#pragma unroll
  for (int b = 0; b < B; b++) {
    const uint32_t bb = (uint32_t)b < nq ? (uint32_t)b : nq - 1;  // unused slots repeat the last query, never stored
#pragma unroll
    for (int i = 0; i < ITERS; i++) q[b][i] = queries[(size_t)bb * qstride16 + lane + i * G];
  }
  __syncthreads();
  const uint32_t n = row_end - row_begin;
  const uint32_t rows_per_step = GPB * U;
  const uint32_t n_tiles = (n + rows_per_step - 1) / rows_per_step;
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t r0 = row_begin + tile * rows_per_step + grp;
    u4 x[U][ITERS];
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t r = r0 + u * GPB;
      if (r >= row_end) r = row_end - 1;  // clamp: recomputed, never stored
      const u4 *p = rows + (size_t)r * stride16;
#pragma unroll
      for (int i = 0; i < ITERS; i++) x[u][i] = load16<true>(p + lane + i * G);
    }
    int v[V];
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        int acc = 0;
#pragma unroll
        for (int i = 0; i < ITERS; i++) {
          acc = __builtin_amdgcn_sdot4((int)x[u][i].x, (int)q[b][i].x, acc, false);
          acc = __builtin_amdgcn_sdot4((int)x[u][i].y, (int)q[b][i].y, acc, false);
          acc = __builtin_amdgcn_sdot4((int)x[u][i].z, (int)q[b][i].z, acc, false);
          acc = __builtin_amdgcn_sdot4((int)x[u][i].w, (int)q[b][i].w, acc, false);
        }
        v[b * U + u] = acc;
      }
    MqRed<G / 2, V, int>::run(v, lane);
#pragma unroll
    for (int p = 0; p < CNT; p++) {
      const uint32_t j = lane * CNT + p, b = j / U, u = j % U;
      const uint32_t r = r0 + u * GPB;
      if (b < nq && r < row_end) {
        const float2 qq = qxs[b];
        const u4 qxv = (u4){0u, __float_as_uint(qq.x), __float_as_uint(qq.y), 0u};
        keys[(size_t)b * keys_ld + r] = to_key(shadow8_distance<METRIC>((float)v[p], row_meta[r], qxv));
      }
    }
  }
}

// ---- INT8 / UINT8 and FLOAT64 rows (round 3: the rest of the element types) --------------------------------------------
// Integer rows: the sums of v_dot4 products are exact integers whatever the order, so the keys are scan_kernel's as soon as
// the SAME integers go through the SAME finish<>().  A row needs sum x.q per query and -- L2, cosine -- sum x.x once; the
// query's extra chunk {sum q^2, |q|} (FlatIndex::upload_query) sits behind every padded query and is kept in LDS.
template <int G>
__device__ __forceinline__ int group_reduce_i(int v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <int TYPE, int METRIC, int G, int ITERS, int U, int B, bool EXACT>
__global__ __launch_bounds__(256, 2) void scan_mq_int_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t chunks,
                                                             uint32_t row_begin, uint32_t row_end,
                                                             const u4 *__restrict__ queries, uint32_t qstride16, uint32_t nq,
                                                             uint32_t *__restrict__ keys, uint32_t keys_ld) {
  constexpr int GPB = 256 / G;
  constexpr int V = U * B, LG = ilog2(G), LV = ilog2(V);
  constexpr int H = LV < LG ? LV : LG;
  constexpr int CNT = V >> H;
  constexpr bool NEED_XX = METRIC == KM_L2 || METRIC == KM_COS;
  __shared__ u4 qxs[B];
  const uint32_t lane = threadIdx.x % G;
  const uint32_t grp = threadIdx.x / G;
  if (threadIdx.x < B) qxs[threadIdx.x] = queries[(size_t)(threadIdx.x < nq ? threadIdx.x : nq - 1) * qstride16 + chunks];
  u4 q[B][ITERS];
#pragma unroll
  for (int b = 0; b < B; b++) {
    const uint32_t bb = (uint32_t)b < nq ? (uint32_t)b : nq - 1;  // unused slots repeat the last query, never stored
#pragma unroll
    for (int i = 0; i < ITERS; i++) q[b][i] = load_chunk<EXACT, false>(queries + (size_t)bb * qstride16, lane + i * G, chunks);
  }
  __syncthreads();

  constexpr bool INTERLEAVE = G < 64;  // (row <-> group mapping of scan_kernel)
  const uint32_t n = row_end - row_begin;
  const uint32_t rows_per_step = INTERLEAVE ? GPB * U : U;
  const uint32_t n_tiles = (n + rows_per_step - 1) / rows_per_step;
  const uint32_t tile0 = INTERLEAVE ? blockIdx.x : blockIdx.x * GPB + grp;
  const uint32_t tile_step = INTERLEAVE ? gridDim.x : gridDim.x * GPB;
  const uint32_t u_stride = INTERLEAVE ? GPB : 1;
  const uint32_t jtop = lane >> (LG - H);
  const bool writer = (lane & ((1u << (LG - H)) - 1u)) == 0;

  for (uint32_t tile = tile0; tile < n_tiles; tile += tile_step) {
    const uint32_t r0 = row_begin + tile * rows_per_step + (INTERLEAVE ? grp : 0);
    u4 x[U][ITERS];
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t r = r0 + u * u_stride;
      if (r >= row_end) r = row_end - 1;  // clamp: recomputed, never stored
      const u4 *p = rows + (size_t)r * stride16;
#pragma unroll
      for (int i = 0; i < ITERS; i++) x[u][i] = load_chunk<EXACT, true>(p, lane + i * G, chunks);
    }
    int v[V], xx[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
      for (int b = 0; b < B; b++) {
        int acc = 0;
#pragma unroll
        for (int i = 0; i < ITERS; i++) {
          acc = dot4<TYPE>(x[u][i].x, q[b][i].x, acc);
          acc = dot4<TYPE>(x[u][i].y, q[b][i].y, acc);
          acc = dot4<TYPE>(x[u][i].z, q[b][i].z, acc);
          acc = dot4<TYPE>(x[u][i].w, q[b][i].w, acc);
        }
        v[b * U + u] = acc;
      }
      int a = 0;
      if constexpr (NEED_XX) {
#pragma unroll
        for (int i = 0; i < ITERS; i++) {
          a = dot4<TYPE>(x[u][i].x, x[u][i].x, a);
          a = dot4<TYPE>(x[u][i].y, x[u][i].y, a);
          a = dot4<TYPE>(x[u][i].z, x[u][i].z, a);
          a = dot4<TYPE>(x[u][i].w, x[u][i].w, a);
        }
        a = group_reduce_i<G>(a);
      }
      xx[u] = a;
    }
    if constexpr (G > 1) MqRed<G / 2, V, int>::run(v, lane);
#pragma unroll
    for (int p = 0; p < CNT; p++) {
      const uint32_t j = jtop * CNT + p, b = j / U, u = j % U;
      const uint32_t r = r0 + u * u_stride;
      int xxu = xx[0];
#pragma unroll
      for (int uu = 1; uu < U; uu++) xxu = u == (uint32_t)uu ? xx[uu] : xxu;  // (no dynamic register indexing)
      if (writer && b < nq && r < row_end) keys[(size_t)b * keys_ld + r] = to_key(finish<TYPE, METRIC>(I2{v[p], xxu}, qxs[b]));
    }
  }
}

// FLOAT64 rows: fp64 partial sums in scan_kernel's per-lane order, the halving butterfly on doubles (plain shuffles: the
// 32-bit lane swaps do not apply), 64-bit keys -- one array of keys_ld u64 per query.
template <int METRIC, int G, int ITERS, int U, int B, bool EXACT>
__global__ __launch_bounds__(256, 2) void scan_mq_f64_kernel(const u4 *__restrict__ rows, uint32_t stride16, uint32_t chunks,
                                                             uint32_t row_begin, uint32_t row_end,
                                                             const u4 *__restrict__ queries, uint32_t qstride16, uint32_t nq,
                                                             uint64_t *__restrict__ keys, uint32_t keys_ld) {
  constexpr int GPB = 256 / G;
  constexpr int V = U * B, LG = ilog2(G), LV = ilog2(V);
  constexpr int H = LV < LG ? LV : LG;
  constexpr int CNT = V >> H;
  const uint32_t lane = threadIdx.x % G;
  const uint32_t grp = threadIdx.x / G;
  u4 q[B][ITERS];
#pragma unroll
  for (int b = 0; b < B; b++) {
    const uint32_t bb = (uint32_t)b < nq ? (uint32_t)b : nq - 1;
#pragma unroll
    for (int i = 0; i < ITERS; i++) q[b][i] = load_chunk<EXACT, false>(queries + (size_t)bb * qstride16, lane + i * G, chunks);
  }
  constexpr bool INTERLEAVE = G < 64;
  const uint32_t n = row_end - row_begin;
  const uint32_t rows_per_step = INTERLEAVE ? GPB * U : U;
  const uint32_t n_tiles = (n + rows_per_step - 1) / rows_per_step;
  const uint32_t tile0 = INTERLEAVE ? blockIdx.x : blockIdx.x * GPB + grp;
  const uint32_t tile_step = INTERLEAVE ? gridDim.x : gridDim.x * GPB;
  const uint32_t u_stride = INTERLEAVE ? GPB : 1;
  const uint32_t jtop = lane >> (LG - H);
  const bool writer = (lane & ((1u << (LG - H)) - 1u)) == 0;

  for (uint32_t tile = tile0; tile < n_tiles; tile += tile_step) {
    const uint32_t r0 = row_begin + tile * rows_per_step + (INTERLEAVE ? grp : 0);
    u4 x[U][ITERS];
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t r = r0 + u * u_stride;
      if (r >= row_end) r = row_end - 1;
      const u4 *p = rows + (size_t)r * stride16;
#pragma unroll
      for (int i = 0; i < ITERS; i++) x[u][i] = load_chunk<EXACT, true>(p, lane + i * G, chunks);
    }
    double v[V];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int b = 0; b < B; b++) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < ITERS; i++) acc = Op<KT_F64, METRIC>::add(acc, x[u][i], q[b][i]);
        v[b * U + u] = acc;
      }
    if constexpr (G > 1) MqRed<G / 2, V, double>::run(v, lane);
#pragma unroll
    for (int p = 0; p < CNT; p++) {
      const uint32_t j = jtop * CNT + p, b = j / U, u = j % U;
      const uint32_t r = r0 + u * u_stride;
      if (writer && b < nq && r < row_end) keys[(size_t)b * keys_ld + r] = to_key(finish<KT_F64, METRIC>(v[p], zero4()));
    }
  }
}

std::atomic<uint64_t> g_last_mq{0};

struct MqCtx {
  const u4 *rows;
  uint32_t stride16, chunks, row_begin, row_end;
  const u4 *queries;
  uint32_t qstride16, nq;
  uint32_t *keys;
  uint32_t keys_ld;
  hipStream_t s;
};

template <int TYPE, int METRIC, int G, int ITERS, int U, int B>
void mq_launch_one(const MqCtx &c) {
  const ScanTuning &t = scan_tuning();
  constexpr int GPB = 256 / G;
  const uint32_t n = c.row_end - c.row_begin;
  const bool exact = c.chunks == (uint32_t)(G * ITERS);
  const uint32_t need = G < 64 ? (n + GPB * U - 1) / (GPB * U) : ((n + U - 1) / U + GPB - 1) / GPB;
  // grid cap per CU, measured at 10 M x 768 fp32 (profiles/r03_mq_scan_ab.txt): four-query kernel 4 (4.68 ms; 2: 5.7-6.1,
  // 8: 4.74, 16: 4.95), eight-query kernel 8 (5.07 ms; 2: 5.14, 4: 5.31, 16: 5.09)
  const uint32_t cap = (uint32_t)(t.num_cus * (t.mq_blocks_per_cu > 0 ? t.mq_blocks_per_cu : (B <= 4 ? 4 : 8)));  // (B = 16: 8, two resident)
  const uint32_t grid = need < cap ? need : cap;
  if (!grid) return;
  g_last_mq = (uint64_t)TYPE | ((uint64_t)METRIC << 3) | ((uint64_t)G << 6) | ((uint64_t)ITERS << 13) | ((uint64_t)U << 17) |
              ((uint64_t)B << 21) | ((uint64_t)exact << 26) | ((uint64_t)grid << 41);
  if (exact)
    hipLaunchKernelGGL((scan_mq_kernel<TYPE, METRIC, G, ITERS, U, B, true>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16,
                       c.chunks, c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, c.keys, c.keys_ld);
  else
    hipLaunchKernelGGL((scan_mq_kernel<TYPE, METRIC, G, ITERS, U, B, false>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16,
                       c.chunks, c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, c.keys, c.keys_ld);
}

template <int METRIC, int G, int ITERS, int U>
void mq_launch_16(const MqCtx &c) {
  const ScanTuning &t = scan_tuning();
  constexpr int GPB = 256 / G;
  const uint32_t n = c.row_end - c.row_begin;
  const bool exact = c.chunks == (uint32_t)(G * ITERS);
  const uint32_t need = G < 64 ? (n + GPB * U - 1) / (GPB * U) : ((n + U - 1) / U + GPB - 1) / GPB;
  const uint32_t cap = (uint32_t)(t.num_cus * (t.mq_blocks_per_cu > 0 ? t.mq_blocks_per_cu : 8));
  const uint32_t grid = need < cap ? need : cap;
  if (!grid) return;
  g_last_mq = (uint64_t)KT_F32 | ((uint64_t)METRIC << 3) | ((uint64_t)G << 6) | ((uint64_t)ITERS << 13) | ((uint64_t)U << 17) |
              ((uint64_t)16 << 21) | ((uint64_t)exact << 26) | ((uint64_t)1 << 27) | ((uint64_t)grid << 41);
  if (exact)
    hipLaunchKernelGGL((scan_mq16_kernel<METRIC, G, ITERS, U, true>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16, c.chunks,
                       c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, c.keys, c.keys_ld);
  else
    hipLaunchKernelGGL((scan_mq16_kernel<METRIC, G, ITERS, U, false>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16, c.chunks,
                       c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, c.keys, c.keys_ld);
}

template <int TYPE, int METRIC, int G, int ITERS, int U>
void mq_launch_b(const MqCtx &c) {
  // fp16 / bf16 with eight queries: the widening temporaries of three or four chunks do not fit 256 registers next to
  // U rows in flight -- fewer rows per step (register budgets: scripts/kernel_resources.py, tests/test_kernel_resources_cpu.py)
  constexpr int U8 = (TYPE != KT_F32 && ITERS >= 3) ? 2 : U;
  if (c.nq > 8) {
    // nine to sixteen queries.  FLOAT32 rows of up to 3 KiB: ONE pass, sixteen queries in registers (192 of them at dim 768;
    // fewer rows in flight per step).  Everything else: two passes of up to eight.
    if constexpr (TYPE == KT_F32) {
      if (scan_tuning().mq16 == 1) {  // queries in LDS (every FLOAT32 shape: 16 KiB per chunk per lane of query data)
        mq_launch_16<METRIC, G, ITERS, 4>(c);  // (U = 8 spills at three and four chunks per lane)
        return;
      }
      if constexpr (ITERS <= 3) {
        if (scan_tuning().mq16 == 2) {  // queries in registers (A/B: the first form of the sixteen-query pass)
          mq_launch_one<TYPE, METRIC, G, ITERS, (ITERS == 1 ? 4 : (ITERS == 2 ? 2 : kMq16U3)), 16>(c);
          return;
        }
      }
    }
    MqCtx lo = c, hi = c;
    lo.nq = 8;
    hi.nq = c.nq - 8;
    hi.queries = c.queries + 8 * (size_t)c.qstride16;
    hi.keys = c.keys + 8 * (size_t)c.keys_ld;
    mq_launch_b<TYPE, METRIC, G, ITERS, U>(lo);
    mq_launch_b<TYPE, METRIC, G, ITERS, U>(hi);
    return;
  }
  if (c.nq <= 4) {
    mq_launch_one<TYPE, METRIC, G, ITERS, U, 4>(c);
  } else if (TYPE != KT_F32 && ITERS >= 4) {  // 4 KiB fp16 / bf16 rows: eight queries spill -- two passes of four
    MqCtx lo = c, hi = c;
    lo.nq = 4;
    hi.nq = c.nq - 4;
    hi.queries = c.queries + 4 * (size_t)c.qstride16;
    hi.keys = c.keys + 4 * (size_t)c.keys_ld;
    mq_launch_one<TYPE, METRIC, G, ITERS, U, 4>(lo);
    mq_launch_one<TYPE, METRIC, G, ITERS, U, 4>(hi);
  } else {
    mq_launch_one<TYPE, METRIC, G, ITERS, U8, 8>(c);
  }
}

template <int TYPE, int METRIC>
bool mq_launch_shape(const MqCtx &c) {
  const Shape sh = pick_shape(c.chunks);
  if (sh.G == 64) {
    switch (sh.ITERS) {
      case 1: mq_launch_b<TYPE, METRIC, 64, 1, 4>(c); return true;
      case 2: mq_launch_b<TYPE, METRIC, 64, 2, 4>(c); return true;
      case 3: mq_launch_b<TYPE, METRIC, 64, 3, 4>(c); return true;
      case 4: mq_launch_b<TYPE, METRIC, 64, 4, 2>(c); return true;
      default: return false;
    }
  }
  if (sh.G == 32) {
    switch (sh.ITERS) {
      case 1: mq_launch_b<TYPE, METRIC, 32, 1, 4>(c); return true;
      case 3: mq_launch_b<TYPE, METRIC, 32, 3, 4>(c); return true;
      default: return false;
    }
  }
  return false;
}

// ---- launches of the integer / FLOAT64 kernels: eight (four) queries per launch, further queries in further launches ----
template <int G, int U>
uint32_t mq_grid(const MqCtx &c) {
  const ScanTuning &t = scan_tuning();
  constexpr int GPB = 256 / G;
  const uint32_t n = c.row_end - c.row_begin;
  const uint32_t need = G < 64 ? (n + GPB * U - 1) / (GPB * U) : ((n + U - 1) / U + GPB - 1) / GPB;
  const uint32_t cap = (uint32_t)(t.num_cus * (t.mq_blocks_per_cu > 0 ? t.mq_blocks_per_cu : 8));
  return need < cap ? need : cap;
}

template <int TYPE, int METRIC, int G, int ITERS, int U>
void mq_launch_int(const MqCtx &c) {
  constexpr int B = 8;
  if (c.nq > (uint32_t)B) {
    MqCtx lo = c, hi = c;
    lo.nq = B;
    hi.nq = c.nq - B;
    hi.queries = c.queries + B * (size_t)c.qstride16;
    hi.keys = c.keys + B * (size_t)c.keys_ld;
    mq_launch_int<TYPE, METRIC, G, ITERS, U>(lo);
    mq_launch_int<TYPE, METRIC, G, ITERS, U>(hi);
    return;
  }
  const uint32_t grid = mq_grid<G, U>(c);
  if (!grid) return;
  const bool exact = c.chunks == (uint32_t)(G * ITERS);
  g_last_mq = (uint64_t)TYPE | ((uint64_t)METRIC << 3) | ((uint64_t)G << 6) | ((uint64_t)ITERS << 13) | ((uint64_t)U << 17) |
              ((uint64_t)B << 21) | ((uint64_t)exact << 26) | ((uint64_t)2 << 28) | ((uint64_t)grid << 41);
  if (exact)
    hipLaunchKernelGGL((scan_mq_int_kernel<TYPE, METRIC, G, ITERS, U, B, true>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16,
                       c.chunks, c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, c.keys, c.keys_ld);
  else
    hipLaunchKernelGGL((scan_mq_int_kernel<TYPE, METRIC, G, ITERS, U, B, false>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16,
                       c.chunks, c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, c.keys, c.keys_ld);
}

template <int TYPE, int METRIC>
bool mq_launch_shape_int(const MqCtx &c) {
  const Shape sh = pick_shape(c.chunks);
  if (sh.ITERS == 1) {
    switch (sh.G) {
      case 8: mq_launch_int<TYPE, METRIC, 8, 1, 4>(c); return true;
      case 16: mq_launch_int<TYPE, METRIC, 16, 1, 4>(c); return true;
      case 32: mq_launch_int<TYPE, METRIC, 32, 1, 4>(c); return true;
      case 64: mq_launch_int<TYPE, METRIC, 64, 1, 4>(c); return true;
      default: return false;
    }
  }
  if (sh.G == 16 && sh.ITERS == 3) { mq_launch_int<TYPE, METRIC, 16, 3, 4>(c); return true; }
  if (sh.G == 32 && sh.ITERS == 3) { mq_launch_int<TYPE, METRIC, 32, 3, 4>(c); return true; }
  if (sh.G == 64) {
    switch (sh.ITERS) {
      case 2: mq_launch_int<TYPE, METRIC, 64, 2, 4>(c); return true;
      case 3: mq_launch_int<TYPE, METRIC, 64, 3, 2>(c); return true;
      case 4: mq_launch_int<TYPE, METRIC, 64, 4, 2>(c); return true;
      default: return false;
    }
  }
  return false;
}

template <int METRIC, int G, int ITERS, int U>
void mq_launch_f64(const MqCtx &c) {
  constexpr int B = 4;
  if (c.nq > (uint32_t)B) {
    MqCtx lo = c, hi = c;
    lo.nq = B;
    hi.nq = c.nq - B;
    hi.queries = c.queries + B * (size_t)c.qstride16;
    hi.keys = c.keys + 2 * B * (size_t)c.keys_ld;  // (u32 words: the keys are 64 bits wide)
    mq_launch_f64<METRIC, G, ITERS, U>(lo);
    mq_launch_f64<METRIC, G, ITERS, U>(hi);
    return;
  }
  const uint32_t grid = mq_grid<G, U>(c);
  if (!grid) return;
  const bool exact = c.chunks == (uint32_t)(G * ITERS);
  g_last_mq = (uint64_t)KT_F64 | ((uint64_t)METRIC << 3) | ((uint64_t)G << 6) | ((uint64_t)ITERS << 13) | ((uint64_t)U << 17) |
              ((uint64_t)B << 21) | ((uint64_t)exact << 26) | ((uint64_t)3 << 28) | ((uint64_t)grid << 41);
  uint64_t *keys = reinterpret_cast<uint64_t *>(c.keys);
  if (exact)
    hipLaunchKernelGGL((scan_mq_f64_kernel<METRIC, G, ITERS, U, B, true>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16, c.chunks,
                       c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, keys, c.keys_ld);
  else
    hipLaunchKernelGGL((scan_mq_f64_kernel<METRIC, G, ITERS, U, B, false>), dim3(grid), dim3(256), 0, c.s, c.rows, c.stride16, c.chunks,
                       c.row_begin, c.row_end, c.queries, c.qstride16, c.nq, keys, c.keys_ld);
}

template <int METRIC>
bool mq_launch_shape_f64(const MqCtx &c) {
  const Shape sh = pick_shape(c.chunks);
  if (sh.G == 32 && sh.ITERS == 1) { mq_launch_f64<METRIC, 32, 1, 4>(c); return true; }
  if (sh.G == 32 && sh.ITERS == 3) { mq_launch_f64<METRIC, 32, 3, 4>(c); return true; }
  if (sh.G == 64) {
    switch (sh.ITERS) {
      case 1: mq_launch_f64<METRIC, 64, 1, 4>(c); return true;
      case 2: mq_launch_f64<METRIC, 64, 2, 4>(c); return true;
      case 3: mq_launch_f64<METRIC, 64, 3, 2>(c); return true;
      case 4: mq_launch_f64<METRIC, 64, 4, 2>(c); return true;
      case 6: mq_launch_f64<METRIC, 64, 6, 2>(c); return true;
      default: return false;
    }
  }
  return false;
}

// which (G, ITERS) shapes the integer / FLOAT64 kernels are instantiated for (mirrors the two functions above)
bool mq_int_shape(uint32_t stride16) {
  const Shape sh = pick_shape(stride16);
  if (sh.ITERS == 1) return sh.G >= 8 && sh.G <= 64;
  return ((sh.G == 16 || sh.G == 32) && sh.ITERS == 3) || (sh.G == 64 && sh.ITERS >= 2 && sh.ITERS <= 4);
}
bool mq_f64_shape(uint32_t stride16) {
  const Shape sh = pick_shape(stride16);
  if (sh.G == 32) return sh.ITERS == 1 || sh.ITERS == 3;
  return sh.G == 64 && (sh.ITERS == 1 || sh.ITERS == 2 || sh.ITERS == 3 || sh.ITERS == 4 || sh.ITERS == 6);
}

}  // namespace

bool scan_mq_supported(int type, int metric, uint32_t stride16) {
  if (type == KT_I8 || type == KT_U8) return (metric == KM_IP || metric == KM_L2 || metric == KM_COS) && mq_int_shape(stride16);
  if (metric != KM_IP && metric != KM_L2) return false;
  if (type == KT_F64) return mq_f64_shape(stride16);
  if (type != KT_F32 && type != KT_F16 && type != KT_BF16) return false;
  const Shape sh = pick_shape(stride16);
  return (sh.G == 64 && sh.ITERS >= 1 && sh.ITERS <= 4) || (sh.G == 32 && (sh.ITERS == 1 || sh.ITERS == 3));
}

bool launch_scan_mq(const void *rows, size_t stride, int type, int metric, uint32_t row_begin, uint32_t row_end,
                    const void *queries, size_t qstride, uint32_t nq, void *keys, uint32_t keys_ld, hipStream_t s) {
  if (row_end <= row_begin || !nq || nq > kMqMaxQueries || !scan_mq_supported(type, metric, (uint32_t)(stride / 16))) return false;
  const MqCtx c{(const u4 *)rows, (uint32_t)(stride / 16), (uint32_t)(stride / 16), row_begin, row_end,
                (const u4 *)queries, (uint32_t)(qstride / 16), nq, (uint32_t *)keys, keys_ld, s};
#define RSGPU_MQ_CASE(T)                                                \
  case T:                                                               \
    return metric == KM_L2 ? mq_launch_shape<T, KM_L2>(c) : mq_launch_shape<T, KM_IP>(c);
#define RSGPU_MQ_CASE_INT(T)                                            \
  case T:                                                               \
    return metric == KM_L2 ? mq_launch_shape_int<T, KM_L2>(c)           \
                           : (metric == KM_IP ? mq_launch_shape_int<T, KM_IP>(c) : mq_launch_shape_int<T, KM_COS>(c));
  switch (type) {
    RSGPU_MQ_CASE(KT_F32)
    RSGPU_MQ_CASE(KT_F16)
    RSGPU_MQ_CASE(KT_BF16)
    RSGPU_MQ_CASE_INT(KT_I8)
    RSGPU_MQ_CASE_INT(KT_U8)
    case KT_F64: return metric == KM_L2 ? mq_launch_shape_f64<KM_L2>(c) : mq_launch_shape_f64<KM_IP>(c);
    default: return false;
  }
#undef RSGPU_MQ_CASE
#undef RSGPU_MQ_CASE_INT
}

bool scan_mq_i8_supported(uint32_t stride16) { return stride16 == 16 || stride16 == 32 || stride16 == 48 || stride16 == 64; }

bool launch_scan_mq_i8(const void *rows, size_t stride, int metric, uint32_t row_begin, uint32_t row_end, const float *row_meta,
                       const void *queries, size_t qstride, const float *qx, uint32_t nq, uint32_t *keys, uint32_t keys_ld,
                       hipStream_t s) {
  const uint32_t s16 = (uint32_t)(stride / 16);
  if (row_end <= row_begin || !nq || nq > 8 || !scan_mq_i8_supported(s16) || (metric != KM_IPS && metric != KM_L2S)) return false;
  constexpr int U = 4, B = 8;
  const uint32_t n = row_end - row_begin, need = (n + 16 * U - 1) / (16 * U);
  const ScanTuning &t = scan_tuning();
  const uint32_t cap = (uint32_t)(t.num_cus * (t.mq_blocks_per_cu > 0 ? t.mq_blocks_per_cu : 8));
  const dim3 grid(need < cap ? need : cap), block(256);
#define RSGPU_MQI8(MM, II)                                                                                                 \
  hipLaunchKernelGGL((scan_mq_i8_kernel<MM, II, U, B>), grid, block, 0, s, (const u4 *)rows, s16, row_begin, row_end,     \
                     (const float2 *)row_meta, (const u4 *)queries, (uint32_t)(qstride / 16), (const float2 *)qx, nq, keys,  \
                     keys_ld)
#define RSGPU_MQI8_M(II)                       \
  do {                                         \
    if (metric == KM_L2S) RSGPU_MQI8(KM_L2S, II); \
    else RSGPU_MQI8(KM_IPS, II);               \
  } while (0)
  switch (s16 / 16) {
    case 1: RSGPU_MQI8_M(1); break;
    case 2: RSGPU_MQI8_M(2); break;
    case 3: RSGPU_MQI8_M(3); break;
    default: RSGPU_MQI8_M(4); break;
  }
#undef RSGPU_MQI8_M
#undef RSGPU_MQI8
  return true;
}

const char *last_scan_mq_kernel_name(char *buf, size_t cap) {
  static const char *tn[] = {"f32", "f64", "bf16", "f16", "i8", "u8"}, *mn[] = {"L2", "IP", "COS", "IPS", "L2S", "?", "?", "?"};
  const uint64_t v = g_last_mq.load();
  if (!v) {
    snprintf(buf, cap, "none");
    return buf;
  }
  const unsigned fam = (unsigned)((v >> 28) & 3);
  snprintf(buf, cap, "%s<%s,%s,G=%u,ITERS=%u,U=%u,B=%u,EXACT=%u> grid=%ux256",
           fam == 2 ? "scan_mq_int_kernel" : (fam == 3 ? "scan_mq_f64_kernel" : (((v >> 27) & 1) ? "scan_mq16_kernel" : "scan_mq_kernel")),
           tn[v & 7], mn[(v >> 3) & 7],
           (unsigned)((v >> 6) & 127), (unsigned)((v >> 13) & 15), (unsigned)((v >> 17) & 15), (unsigned)((v >> 21) & 31),
           (unsigned)((v >> 26) & 1), (unsigned)(v >> 41));
  return buf;
}

}  // namespace rsgpu
