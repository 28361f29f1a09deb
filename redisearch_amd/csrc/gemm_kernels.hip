// gemm_kernels.hip -- batched-query FLAT KNN on the matrix cores (gfx950 MFMA), hand-written HIP.
//
// The reference has no batched API: B queries are B independent VecSimIndex_TopKQuery calls
// (reference src/iterators/hybrid_reader.c:374), each streaming the whole corpus.  With B = 256 fp16
// queries the same corpus pass becomes a true GEMM  S[256 x N] = Q[256 x dim] * X^T  with arithmetic
// intensity 256 flop/B -- at the MFMA/HBM ridge of MI355X -- so THIS is where MFMA belongs
// (BASELINE configs[2]; the single-query path in scan_kernels.hip stays a bandwidth-bound GEMV).
//
// S is never materialised (256 x 10M fp32 = 10 GB).  Three steps, all on the device:
//   1. the GEMM in KEYS mode over a small prefix of the corpus (n0 rows) writes every orderable
//      distance key; batch_select_kernel finds, per query, the k-th smallest key of the sample: a valid
//      upper bound tau[q] of the final k-th distance;
//   2. the GEMM in FILTER mode over ALL rows keeps only (row,key) with distance <= tau[q] -- expected
//      k*N/n0 survivors per query -- appended to per-query candidate lists;
//   3. batch_select_kernel picks, per query, the exact k smallest (key,row) composites of its
//      candidates (same total order as select_kernels.hip).
// The result is exact whatever the data order (tau is a true upper bound); a candidate overflow makes
// the host redo that query on the single-query path.
//
// GEMM: 512 threads = 8 wavefronts (4 along the queries x 2 along the corpus rows), block tile
// 256 queries x 128 rows, v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulate; each wave owns a 64 x 64
// sub-tile = 2 x 2 MFMA tiles = 64 accumulator VGPRs (a 256 x 256 tile with 128 accumulators per lane
// spilled ~100 VGPRs under hipcc).  Both operands are K-contiguous rows (an "NT" GEMM).  Two staging
// schemes:
//   * gemm_topk_ring_kernel: global -> LDS directly (global_load_lds_dwordx4, 1 KiB per
//     wave-instruction, no VGPRs, no ds_write pass) into a ring of three stages, two in flight while the
//     third is multiplied; counted s_waitcnt vmcnt(N) + raw s_barrier keep the DMAs alive across the
//     barrier (cdna_hip_programming.md "Pipelining across barriers").  The DMA destination is
//     lane-linear, so the XOR swizzle is applied to the SOURCE chunk each lane fetches and again on the
//     fragment read (same involution on both sides).  KC = 8 chunks per stage (64 halves, 48 KiB stage,
//     one workgroup per CU) or KC = 4 (24 KiB stage, two workgroups per CU so that one multiplies while
//     the other waits).  Needs dim % (8*KC) == 0.
//   * gemm_topk_kernel: global -> registers -> LDS, double-buffered, any K (zero-filled tail).
// The epilogue is specialised per mode and its rare append path is out of line, so that it does not
// inflate the register allocation of the K loop.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace rsgpu {
namespace {

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int BM = 256, BN = 128;

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

// XOR swizzle of the 16-byte chunk index inside an LDS row of KC chunks: conflict-free for the
// 16-lane groups ds_read_b128 is serviced in (rows l, same chunk): KC=8 -> 128-byte rows, key
// (row>>1)&7; KC=4 -> 64-byte rows, key (row>>2)&3.
template <int KC>
__device__ __forceinline__ uint32_t swz(uint32_t row) {
  return KC == 8 ? ((row >> 1) & 7u) : ((row >> 2) & 3u);
}
template <int KC>
__device__ __forceinline__ uint32_t slotk(uint32_t row, uint32_t c) { return row * KC + (c ^ swz<KC>(row)); }

struct GemmArgs {
  const u4 *rows;      // corpus, row-contiguous
  const u4 *queries;   // [256][stride16], zero padded
  uint32_t stride16;   // row stride in 16-byte chunks == K chunks
  uint32_t row_begin, row_end;
  uint32_t *keys_out;  // KEYS mode: [256][keys_ld], column = row - row_begin
  uint32_t keys_ld;
  const float *tau;    // FILTER mode: per-query distance upper bound
  uint32_t *cand_count;  // [256]
  uint2 *cand;           // [256][cand_cap] (row, key)
  uint32_t cand_cap;
};

__device__ __noinline__ void append_candidate(uint32_t *cand_count, uint2 *cand, uint32_t cand_cap, uint32_t q,
                                              uint32_t row, float d) {
  uint32_t s = atomicAdd(&cand_count[q], 1u);
  if (s < cand_cap) cand[(size_t)q * cand_cap + s] = make_uint2(row, f2key(d));
}

// C row = query (reg&3)+8*(reg>>2)+4*(lane>>5) of the 32x32 MFMA tile, C col = corpus row lane&31.
// MODE 0: write every key; MODE 1: keep distance <= tau[q].
template <int MODE>
__device__ __forceinline__ void epilogue(const GemmArgs &g, const f32x16 (&acc)[2][2], const float *tau_s,
                                         uint32_t row0, uint32_t wm, uint32_t wn, uint32_t lane) {
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const uint32_t xr = row0 + wn * 64 + nt * 32 + (lane & 31);
      const bool live = xr < g.row_end;
      const uint32_t qb = wm * 64 + mt * 32 + 4 * (lane >> 5);
      if (MODE == 0) {
        uint32_t *out = g.keys_out + (size_t)qb * g.keys_ld + (xr - g.row_begin);
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (live) out[(size_t)((r & 3) + 8 * (r >> 2)) * g.keys_ld] = f2key(1.0f - acc[mt][nt][r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const uint32_t q = qb + (r & 3) + 8 * (r >> 2);
          const float d = 1.0f - acc[mt][nt][r];
          if (live && d <= tau_s[q]) append_candidate(g.cand_count, g.cand, g.cand_cap, q, xr, d);
        }
      }
    }
  }
}

// counted wait: at most n VMEM operations of this wave still outstanding (n must become an immediate)
__device__ __forceinline__ void wait_vmcnt(uint32_t n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// ---- LDS-DMA ring ---------------------------------------------------------------------------------------
// NS stages in the ring: NS-1 in flight while one is multiplied.
template <int DT, int KC, int NS, int MINW, int MODE>
__global__ __launch_bounds__(512, MINW) void gemm_topk_ring_kernel(GemmArgs g) {
  constexpr int NSTAGE = NS;
  constexpr int STAGE = (BM + BN) * KC;    // chunks per stage
  constexpr int RPP = 64 / KC;             // rows per 64-chunk DMA piece
  constexpr int PIECES = (BM + BN) / RPP;  // pieces per stage (48 or 24)
  constexpr int PPW = PIECES / 8;          // pieces per wave (6 or 3)
  constexpr int A_PIECES = BM / RPP;
  __shared__ u4 smem[NSTAGE * STAGE + BM / 4];  // ring + tau[256]: ONE __shared__ object (a second one makes
                                                // hipcc drain vmcnt before every ds_read)
  float *tau_s = reinterpret_cast<float *>(smem + NSTAGE * STAGE);
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t wm = w >> 1, wn = w & 1;  // 4 waves along the queries x 2 along the rows
  const uint32_t kchunks = g.stride16;
  const uint32_t ksteps = kchunks / KC;
  const uint32_t n = g.row_end - g.row_begin;
  const uint32_t n_tiles = (n + BN - 1) / BN;
  if (MODE == 1 && tid < BM) tau_s[tid] = g.tau[tid];
  __syncthreads();

  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t row0 = g.row_begin + tile * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.0f;

    // DMA map: a stage is [384 rows][KC chunks] (256 query rows, then 128 corpus rows) cut in pieces of
    // 64 chunks; wave w moves pieces PPW*w ...  Lane l of piece j lands in slot 64j + l = row
    // RPP*j + l/KC, slot l%KC, and fetches source chunk (l%KC) ^ swz(row).
    const u4 *src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      const uint32_t j = w * PPW + i;
      const uint32_t r = RPP * j + lane / KC;
      const uint32_t c = (lane % KC) ^ swz<KC>(r);
      if (j < A_PIECES) {
        src[i] = g.queries + (size_t)r * kchunks + c;
      } else {
        uint32_t xr = row0 + (r - BM);
        if (xr >= g.row_end) xr = g.row_end - 1;  // tail tile: recomputed, never emitted
        src[i] = g.rows + (size_t)xr * kchunks + c;
      }
    }
    auto issue = [&](uint32_t ks) {
      u4 *stage = smem + (ks % NSTAGE) * STAGE;
#pragma unroll
      for (int i = 0; i < PPW; i++) {
        const uint32_t j = w * PPW + i;
        const __attribute__((address_space(1))) void *gp =
            (const __attribute__((address_space(1))) void *)(src[i] + (size_t)ks * KC);
        __attribute__((address_space(3))) void *lp = (__attribute__((address_space(3))) void *)(stage + j * 64);
        if (j < A_PIECES) __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 2);  // nt: corpus lines are used once
      }
    };

#pragma unroll
    for (int p = 0; p < NSTAGE - 1; p++)
      if ((uint32_t)p < ksteps) issue(p);
    for (uint32_t ks = 0; ks < ksteps; ks++) {
      // stage ks has landed once only this wave's DMAs of the younger stages are outstanding ...
      const uint32_t younger = ksteps - 1 - ks < (uint32_t)(NSTAGE - 2) ? ksteps - 1 - ks : (uint32_t)(NSTAGE - 2);
      wait_vmcnt(younger * PPW);
      // ... for every wave; the barrier also says stage ks-1 has been consumed, freeing its slot
      __builtin_amdgcn_s_barrier();
      if (ks + NSTAGE - 1 < ksteps) issue(ks + NSTAGE - 1);
      const u4 *A = smem + (ks % NSTAGE) * STAGE, *B = A + BM * KC;
#pragma unroll
      for (int kk = 0; kk < KC / 2; kk++) {
        const uint32_t c = kk * 2 + (lane >> 5);
        u4 a[2], b[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) a[mt] = A[slotk<KC>(wm * 64 + mt * 32 + (lane & 31), c)];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) b[nt] = B[slotk<KC>(wn * 64 + nt * 32 + (lane & 31), c)];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) acc[mt][nt] = mfma<DT>(a[mt], b[nt], acc[mt][nt]);
      }
    }
    epilogue<MODE>(g, acc, tau_s, row0, wm, wn, lane);
    // stores/atomics of the epilogue share the vmcnt queue with the next tile's DMAs: drain them, and
    // make sure every wave is done reading the ring before it is refilled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// ---- register-staged fallback (any K) ---------------------------------------------------------------------
template <int DT, int MODE>
__global__ __launch_bounds__(512) void gemm_topk_kernel(GemmArgs g) {
  constexpr int KC = 8;
  constexpr int A_CHUNKS = BM * KC, STAGE = (BM + BN) * KC;
  __shared__ u4 smem[2 * STAGE + BM / 4];
  float *tau_s = reinterpret_cast<float *>(smem + 2 * STAGE);
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t wm = w >> 1, wn = w & 1;
  const uint32_t kchunks = g.stride16;
  const uint32_t ksteps = (kchunks + KC - 1) / KC;
  const uint32_t n = g.row_end - g.row_begin;
  const uint32_t n_tiles = (n + BN - 1) / BN;
  if (MODE == 1 && tid < BM) tau_s[tid] = g.tau[tid];
  __syncthreads();
  // staging map: chunks id = tid + i*512 of the query tile (i<4) and of the corpus tile (i<2): row id/8,
  // chunk id%8 -- 8 consecutive lanes cover one 128-byte line
  const uint32_t schunk = tid & 7u, srow0 = tid >> 3;

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
      const uint32_t c = ks * KC + schunk;
      const bool in_k = c < kchunks;
#pragma unroll
      for (int i = 0; i < 4; i++)
        ra[i] = in_k ? g.queries[(size_t)(srow0 + 64 * i) * kchunks + c] : (u4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 2; i++) {
        uint32_t xr = row0 + srow0 + 64 * i;
        if (xr >= g.row_end) xr = g.row_end - 1;
        rb[i] = in_k ? __builtin_nontemporal_load(g.rows + (size_t)xr * kchunks + c) : (u4){0u, 0u, 0u, 0u};
      }
    };
    auto store_stage = [&](uint32_t buf) {
      u4 *A = smem + buf * STAGE, *B = A + A_CHUNKS;
#pragma unroll
      for (int i = 0; i < 4; i++) A[slotk<KC>(srow0 + 64 * i, schunk)] = ra[i];
#pragma unroll
      for (int i = 0; i < 2; i++) B[slotk<KC>(srow0 + 64 * i, schunk)] = rb[i];
    };
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (uint32_t ks = 0; ks < ksteps; ks++) {
      const uint32_t buf = ks & 1u;
      if (ks + 1 < ksteps) load_stage(ks + 1);  // in flight during the MFMAs below
      const u4 *A = smem + buf * STAGE, *B = A + A_CHUNKS;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const uint32_t c = kk * 2 + (lane >> 5);
        u4 a[2], b[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) a[mt] = A[slotk<KC>(wm * 64 + mt * 32 + (lane & 31), c)];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) b[nt] = B[slotk<KC>(wn * 64 + nt * 32 + (lane & 31), c)];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) acc[mt][nt] = mfma<DT>(a[mt], b[nt], acc[mt][nt]);
      }
      if (ks + 1 < ksteps) store_stage(buf ^ 1u);
      __syncthreads();
    }
    epilogue<MODE>(g, acc, tau_s, row0, wm, wn, lane);
    __syncthreads();
  }
}

// ---- per-query block-level radix select ---------------------------------------------------------------
// One workgroup per query.  PAIRS=false: elements are keys[q*ld + i], i<n, composite (key,i);
// PAIRS=true: elements are cand[q*cap + i] = (row,key), i<min(count[q],cap), composite (key,row).
// Finds the k smallest composites; writes the k-th smallest key as a float threshold (tau_out) and/or
// the winners -- in (key, row) order up to 1 024 of them -- to out_rows/out_keys[q*k_ld + j], out_n[q] = how many.
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
  uint32_t stride;     // PAIRS=false: element i is keys[q*ld + i*stride] (strided sample of a key array)
  float slack;         // added to the bound written to tau_out (error band of a low-precision filter pass)
  const float *slack_q;  // per-query band (overrides slack)
  uint32_t prune;        // PAIRS, threshold only: the list keeps just the candidates at or below the new bound
  uint32_t no_regs;      // the list is re-read from memory in every pass even when the registers would hold it (knob batch_select_regs = 0: the tests' way onto the path lists above 16 Ki entries take)
};
__device__ __forceinline__ uint32_t sel_f2key(float f) {  // (scan_ops.hpp f2key: the orderable image of a distance, NaN last)
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xFFFFFFFFu;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <bool PAIRS>
__global__ __launch_bounds__(1024) void batch_select_kernel(BatchSel s) {
  __shared__ uint32_t hist[256];
  __shared__ u64 sh_prefix;
  __shared__ u64 win[1024];
  __shared__ uint32_t sh_krem, sh_exact, sh_levels, sh_out, sh_thr;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  uint32_t n = s.n;
  bool overflowed = false;
  if (PAIRS) {
    uint32_t c = s.cand_count[q];
    if (c > s.cand_cap) {
      if (tid == 0) s.overflow[q] = 1;
      c = s.cand_cap;
      overflowed = true;
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
    return ((u64)s.keys[(size_t)q * s.ld + (size_t)i * s.stride] << 32) | i;
  };
  // The elements stay in registers when 16 per thread hold them (round 6: every digit pass re-read the list from memory, one
  // dependent round trip per 1 024 elements and pass -- 20 us per select at 4 k candidates, 85 us at 16 k, six to seven selects
  // per corpus pass): ONE batch of independent loads, the passes run on LDS alone.
  constexpr int MAXE = 16;
  const bool cached = n <= (uint32_t)MAXE * 1024u && !s.no_regs;  // (uniform)
  const uint32_t nj = (n + 1023u) / 1024u;
  u64 E[MAXE];
  if (cached) {
#pragma unroll
    for (int j = 0; j < MAXE; j++) {
      const uint32_t i = (uint32_t)j * 1024u + tid;
      E[j] = i < n ? elem(i) : ~0ull;
    }
  }
  // a threshold is the key part alone: four digits settle it (the row digits below would only break ties the bound includes anyway)
  const int passes = s.out_rows ? 8 : 4;
  const bool take_all = (k == n);
  for (int p = 0; p < passes && !sh_exact; p++) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const u64 prefix = sh_prefix;
    const int mshift = 64 - 8 * p, dshift = 56 - 8 * p;
    auto count = [&](bool live, u64 c) {
      // (the leading digits of a list of near-equal distances are the same for every element: one atomic per wavefront then,
      // not 64 serialised on one LDS word)
      const bool in = live && (p == 0 || (c >> mshift) == (prefix >> mshift));
      const uint32_t d = (uint32_t)(c >> dshift) & 0xffu;
      const unsigned long long m = __ballot(in);
      if (!m) return;
      const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, __builtin_ctzll(m));
      if (__ballot(in && d != d0) == 0ull) {
        if (lane == (uint32_t)__builtin_ctzll(m)) atomicAdd(&hist[d0], (uint32_t)__popcll(m));
      } else if (in)
        atomicAdd(&hist[d], 1u);
    };
    if (cached) {
#pragma unroll
      for (int j = 0; j < MAXE; j++) {
        if ((uint32_t)j >= nj) break;
        count(E[j] != ~0ull, E[j]);  // (a slot past the end holds ~0: no composite is)
      }
    } else {
      for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
        const uint32_t i = i0 + tid;
        count(i < n, i < n ? elem(i) : 0ull);
      }
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
    sh_thr = 0xFFFFFFFFu;
    if (q >= s.n_valid) {
      s.tau_out[q] = __uint_as_float(0xff800000u);
    } else if (take_all) {  // fewer than k elements: no (new) bound
      if (!PAIRS) s.tau_out[q] = __uint_as_float(0x7f800000u);
    } else {
      uint32_t u = (kk & 0x80000000u) ? (kk ^ 0x80000000u) : ~kk;
      const float t = __uint_as_float(u) + (s.slack_q ? s.slack_q[q] : s.slack);
      s.tau_out[q] = t;
      sh_thr = sel_f2key(t);
    }
  }
  if (PAIRS && s.prune && s.tau_out && !s.out_rows) {  // (uniform)
    // The list keeps only what the NEW bound admits (round 6).  The candidates of the early phases were collected under loose
    // bounds -- the first phase of an fp32 pass keeps all 16 Ki rows it sees -- and every later threshold, the re-scoring and the
    // final select walked them again (selects of 40-80 us).  The re-scoring's own test is this one (key <= f2key(tau), the bounds
    // only fall), so the same candidates reach it; the order within the list is not kept and nothing depends on it.
    __syncthreads();
    const uint32_t thr = sh_thr;
    if (thr != 0xFFFFFFFFu && !overflowed) {
      uint2 *list = const_cast<uint2 *>(s.cand) + (size_t)q * s.cand_cap;
      auto keep_one = [&](bool live, u64 c) {
        const bool keep = live && (uint32_t)(c >> 32) <= thr;
        const unsigned long long m = __ballot(keep);
        if (!m) return;
        const int leader = __builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == (uint32_t)leader) base = atomicAdd(&sh_out, (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (keep) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)c, (uint32_t)(c >> 32));
      };
      if (cached) {  // (every element was read at the start)
#pragma unroll
        for (int j = 0; j < MAXE; j++) {
          if ((uint32_t)j >= nj) break;
          keep_one(E[j] != ~0ull, E[j]);  // (a slot past the end holds ~0: no composite is)
        }
      } else {
        // in place, a chunk of 1 024 at a time: what a chunk keeps lands at or below the chunk's own start + 1 024, so its reads
        // come first (the barrier) and the next chunk's are never touched
        for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
          const uint32_t i = i0 + tid;
          const u64 c = i < n ? elem(i) : 0ull;
          __syncthreads();
          keep_one(i < n, c);
        }
      }
      __syncthreads();
      if (tid == 0) const_cast<uint32_t *>(s.cand_count)[q] = sh_out;
    }
  }
  if (s.out_rows) {
    auto emit = [&](bool live, u64 c) {
      if (live && k > 0 && c <= hi) {
        uint32_t o = atomicAdd(&sh_out, 1u);
        if (o < s.k_ld && o < 1024u) win[o] = c;
        else if (o < s.k_ld) {  // (k_ld above the sorted list's room: as before, unordered)
          s.out_rows[(size_t)q * s.k_ld + o] = (uint32_t)c;
          s.out_keys[(size_t)q * s.k_ld + o] = (uint32_t)(c >> 32);
        }
      }
    };
    if (cached) {
#pragma unroll
      for (int j = 0; j < MAXE; j++) {
        if ((uint32_t)j >= nj) break;
        emit(E[j] != ~0ull, E[j]);  // (a slot past the end holds ~0: no composite is)
      }
    } else {
      for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
        const uint32_t i = i0 + tid;
        emit(i < n, i < n ? elem(i) : 0ull);
      }
    }
    __syncthreads();
    // the winners in (key, row) order (round 6: the host sorted every reply -- 100 x 256 per pass, a fifth of a pass's time): every
    // winner counts the winners before it; composites are distinct (a row appears once)
    const uint32_t got = sh_out < s.k_ld ? sh_out : s.k_ld, ranked = got < 1024u ? got : 1024u;
    for (uint32_t t = tid; t < ranked; t += blockDim.x) {
      const u64 c = win[t];
      uint32_t rank = 0;
#pragma unroll 8
      for (uint32_t j = 0; j < ranked; j++) rank += win[j] < c ? 1u : 0u;
      s.out_rows[(size_t)q * s.k_ld + rank] = (uint32_t)c;
      s.out_keys[(size_t)q * s.k_ld + rank] = (uint32_t)(c >> 32);
    }
    if (tid == 0) s.out_n[q] = got;
  }
}


template <int DT, int MODE>
static void launch_gemm_mode(const GemmArgs &g, uint32_t n_tiles, hipStream_t s) {
  const uint32_t cus = (uint32_t)scan_tuning().num_cus;
  const int variant = scan_tuning().gemm_dma;  // 0 register-staged; 1 ring KC=8, 1 WG/CU; 2 ring KC=4, 2 WG/CU
  const uint32_t grid1 = n_tiles < cus ? n_tiles : cus, grid2 = n_tiles < 2 * cus ? n_tiles : 2 * cus;
  if (variant == 2 && g.stride16 % 4 == 0)       // 24 KiB stages, ring of 3, two workgroups per CU
    hipLaunchKernelGGL((gemm_topk_ring_kernel<DT, 4, 3, 4, MODE>), dim3(grid2), dim3(512), 0, s, g);
  else if (variant == 3 && g.stride16 % 4 == 0)  // 24 KiB stages, ring of 6 (5 in flight), one workgroup per CU
    hipLaunchKernelGGL((gemm_topk_ring_kernel<DT, 4, 6, 2, MODE>), dim3(grid1), dim3(512), 0, s, g);
  else if (variant >= 1 && g.stride16 % 8 == 0)  // 48 KiB stages, ring of 3
    hipLaunchKernelGGL((gemm_topk_ring_kernel<DT, 8, 3, 2, MODE>), dim3(grid1), dim3(512), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_topk_kernel<DT, MODE>), dim3(grid1), dim3(512), 0, s, g);
}

}  // namespace

void launch_gemm_topk(int dtype, const void *rows, const void *queries, uint32_t stride16, uint32_t row_begin,
                      uint32_t row_end, int mode, uint32_t *keys_out, uint32_t keys_ld, const float *tau,
                      uint32_t *cand_count, void *cand, uint32_t cand_cap, hipStream_t s) {
  if (row_end <= row_begin) return;
  GemmArgs g{(const u4 *)rows, (const u4 *)queries, stride16, row_begin, row_end, keys_out, keys_ld, tau,
             cand_count, (uint2 *)cand, cand_cap};
  const uint32_t n_tiles = (row_end - row_begin + BN - 1) / BN;
  if (dtype == KT_F16) {
    if (mode == 0) launch_gemm_mode<KT_F16, 0>(g, n_tiles, s);
    else launch_gemm_mode<KT_F16, 1>(g, n_tiles, s);
  } else {
    if (mode == 0) launch_gemm_mode<KT_BF16, 0>(g, n_tiles, s);
    else launch_gemm_mode<KT_BF16, 1>(g, n_tiles, s);
  }
}

void launch_batch_threshold(const uint32_t *keys, uint32_t ld, uint32_t n, uint32_t k, uint32_t n_queries,
                            uint32_t n_valid, float *tau_out, hipStream_t s, uint32_t stride, float slack,
                            const float *slack_q) {
  BatchSel b{keys, ld, n, nullptr, nullptr, 0, k, tau_out, nullptr, nullptr, nullptr, 0, nullptr, n_valid, stride, slack, slack_q};
  b.no_regs = scan_tuning().batch_select_regs ? 0u : 1u;
  hipLaunchKernelGGL(batch_select_kernel<false>, dim3(n_queries), dim3(1024), 0, s, b);
}

// tightening: tau[q] = k-th smallest distance among the candidates collected so far (all of them are <= the
// old tau, so the new one can only be smaller; a query with fewer than k candidates keeps its bound)
void launch_batch_threshold_cand(const void *cand, const uint32_t *cand_count, uint32_t cand_cap, uint32_t k,
                                 uint32_t n_queries, uint32_t n_valid, float *tau_inout, uint32_t *overflow,
                                 hipStream_t s, float slack, const float *slack_q, bool prune) {
  BatchSel b{nullptr, 0, 0, (const uint2 *)cand, cand_count, cand_cap, k, tau_inout, nullptr, nullptr, nullptr, 0, overflow,
             n_valid, 1, slack, slack_q, prune ? 1u : 0u};
  b.no_regs = scan_tuning().batch_select_regs ? 0u : 1u;
  hipLaunchKernelGGL(batch_select_kernel<true>, dim3(n_queries), dim3(1024), 0, s, b);
}

void launch_batch_select_keys(const uint32_t *keys, uint32_t ld, uint32_t n, uint32_t k, uint32_t n_queries,
                              uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_n, uint32_t k_ld, hipStream_t s) {
  BatchSel b{keys, ld, n, nullptr, nullptr, 0, k, nullptr, out_rows, out_keys, out_n, k_ld, nullptr, n_queries, 1, 0.0f, nullptr};
  b.no_regs = scan_tuning().batch_select_regs ? 0u : 1u;
  hipLaunchKernelGGL(batch_select_kernel<false>, dim3(n_queries), dim3(1024), 0, s, b);
}

void launch_batch_select_cand(const void *cand, const uint32_t *cand_count, uint32_t cand_cap, uint32_t k,
                              uint32_t n_queries, uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_n,
                              uint32_t k_ld, uint32_t *overflow, hipStream_t s) {
  BatchSel b{nullptr, 0, 0, (const uint2 *)cand, cand_count, cand_cap, k, nullptr, out_rows, out_keys, out_n, k_ld, overflow, n_queries, 1, 0.0f, nullptr};
  b.no_regs = scan_tuning().batch_select_regs ? 0u : 1u;
  hipLaunchKernelGGL(batch_select_kernel<true>, dim3(n_queries), dim3(1024), 0, s, b);
}

}  // namespace rsgpu
