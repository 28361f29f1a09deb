// hybrid_kernels.hip -- a whole hybrid query (BASELINE configs[4]: text filter -> BM25 top-N next to an ad-hoc KNN top-k over
// the documents the filter kept) in TWO launches (gfx950, hand-written HIP).
//
// The staged pipeline (postings_kernels.hip + the gather form of scan_kernel) spends a query in ten kernels of 5-27 us and the
// launch gaps between them (profiles/r03_hybrid_one_pass.txt): probe -> scan -> ordered write -> [host reads the hit count]
// -> score -> threshold -> filter -> fetch  ||  labels -> gather -> top-k.  None of the two answers needs the ordered hit
// LIST -- only the caller that asked for it (hits_out) does.  Without it:
//
//   hybrid_tile_kernel    one workgroup per tile of 1 024 consecutive entries of the driving (shortest) list:
//                           * the probe of intersect_probe_kernel -- the tile's window of every other list staged in LDS,
//                             a binary search per driver -- with the match positions kept in registers;
//                           * the hits compacted into LDS and scored one per lane (score_one: the scorers of score_kernel,
//                             same bits); the tile's top-N by (descending score, ascending doc id) -- every hit counts the
//                             hits that beat it and writes itself to the slot of its rank;
//                           * the hits that have a vector, compacted in LDS, their distances -- Op<> / the reduction tree of
//                             scan_kernel: the bits of the gather -- and the tile's top-k by (distance, doc id);
//                           * per tile and at FIXED slots: hit count, N (score key, doc id), k (distance key, doc id), each
//                             list sorted.  No atomics, no fences, no workgroup reads what another one wrote.
//   hybrid_reduce_kernel  one workgroup of 1 024 per branch: the best of every thread's tiles' FIRST entries, a bound from
//                         their k-th, the lists of the few tiles whose first entry passes it ranked in LDS; the winners --
//                         key, doc id -- and the hit count go to pinned host memory.  No tickets, no fences either.
//
// Exactness: the composites are total orders (ties break the way the staged selections break them: by hit order, which is
// doc-id order), so "top-N of the per-tile top-Ns" IS the top-N.  Compiled with -ffp-contract=off
// like postings_kernels.hip: a score has the bits score_kernel gives it; a distance the bits of scan_kernel<GATHER> (explicit
// fmaf chains in scan_ops.hpp, nothing left to contract).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"
#include "postings_ops.hpp"
#include "search_kernels.hpp"

namespace rsgpu {
namespace {

constexpr int kHybDpt = 4;                        // drivers per thread
constexpr uint32_t kHybTile = 256 * kHybDpt;      // drivers per workgroup
constexpr uint32_t kHybMaxTiles = 16384;

// (descending score, ascending doc id): key = ~d2key(score) ascending, then the doc id in the frame the lists share
struct SKey {
  uint64_t k;
  uint32_t i;
};
__device__ __forceinline__ bool sk_less(const SKey &a, const SKey &b) { return a.k != b.k ? a.k < b.k : a.i < b.i; }
__device__ __forceinline__ bool sk_same(const SKey &a, const SKey &b) { return a.k == b.k && a.i == b.i; }
__device__ __forceinline__ SKey sk_none() { return SKey{~0ull, ~0u}; }  // (a real key is never ~0: ~d2key() of a NaN is 0)
// (component by component: a select between two structs is a select between their ADDRESSES -- scratch memory)
__device__ __forceinline__ SKey sk_min(const SKey &a, const SKey &b) {
  const bool lt = sk_less(a, b);
  return SKey{lt ? a.k : b.k, lt ? a.i : b.i};
}
__device__ __forceinline__ float group_reduce_rt(float v, int G) {  // group_reduce<G> with G at run time: the same tree
  for (int m = G >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// The k smallest of a tile's n entries (n <= TILE; entry e = j * 256 + thread sits in mk[j] / mx[j], "none" past n; get(e)
// reads it back from LDS), each handed to put(rank, entry).  Called by the whole workgroup.
//   n <= 192: every entry counts the entries that precede it (n LDS reads each).
//   more (dense tiles -- a one-term filter, lists that are nearly equal): counting all against all is n^2 / 256 reads per lane
//   (0.5 ms per query with every driver a hit).  Instead: the best entry of every lane, the k-th of each wavefront's 64 bests --
//   the smallest of the four bounds the k-th of all from above -- and only the entries at or below it (a few dozen) are
//   ranked.  More of those than the scratch list holds: all against all after all.
constexpr uint32_t kHybDirect = 192, kHybScratch = 256;
template <int DPT, typename Get, typename Put>
__device__ __forceinline__ void tile_select(const uint64_t (&mk)[DPT], const uint32_t (&mx)[DPT], uint32_t n, uint32_t k, Get get,
                                            uint64_t *sk, uint32_t *sx, uint32_t *cnt, uint64_t *wtk, uint32_t *wtx, Put put, bool split) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  bool direct = n <= kHybDirect;
  if (split && n <= 128) {
    // up to 128 entries (the usual tile: a hundred hits): the entries sit in the first n threads and the other wavefronts idled while
    // those counted all n (3.6 of a light tile's 17 us).  The workgroup's 256 threads are W columns (W = 32 / 64 / 128 >= n) x P rows:
    // thread (row, e) counts the entries of the row's SHARE that precede entry e, the shares add up in LDS (sx: free on this path).
    const uint32_t W = n <= 32 ? 32u : (n <= 64 ? 64u : 128u), P = 256u / W;
    const uint32_t e = threadIdx.x & (W - 1), part = threadIdx.x / W, chunk = (n + P - 1) / P;
    const uint32_t lo = part * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (threadIdx.x < W) sx[threadIdx.x] = 0;
    __syncthreads();
    if (e < n && lo < hi) {
      const SKey my = get(e);
      uint32_t rank = 0;
#pragma unroll 8
      for (uint32_t o = lo; o < hi; o++) rank += sk_less(get(o), my) ? 1u : 0u;
      if (rank) atomicAdd(&sx[e], rank);
    }
    __syncthreads();
    if (threadIdx.x < n) {  // (entry e = thread e of slot 0)
      const SKey my{mk[0], mx[0]};
      const uint32_t rank = sx[threadIdx.x];
      if (!sk_same(my, sk_none()) && rank < k) put(rank, my);
    }
    return;
  }
  if (!direct) {
    SKey best = sk_none();
#pragma unroll
    for (int j = 0; j < DPT; j++) best = sk_min(SKey{mk[j], mx[j]}, best);
    sk[threadIdx.x] = best.k;
    sx[threadIdx.x] = best.i;
    if (threadIdx.x == 0) *cnt = 0;
    if (lane == 0) {
      wtk[w] = ~0ull;
      wtx[w] = ~0u;
    }
    __syncthreads();
    if (!sk_same(best, sk_none())) {
      uint32_t rank = 0;
#pragma unroll 8
      for (uint32_t j = 0; j < 64; j++) rank += sk_less(SKey{sk[w * 64 + j], sx[w * 64 + j]}, best) ? 1u : 0u;
      if (rank == k - 1) {
        wtk[w] = best.k;
        wtx[w] = best.i;
      }
    }
    __syncthreads();
    SKey tau = sk_none();
#pragma unroll
    for (int j = 0; j < 4; j++) tau = sk_min(SKey{wtk[j], wtx[j]}, tau);
    __syncthreads();  // (sk / sx are rewritten)
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const SKey c{mk[j], mx[j]};
      if (!sk_same(c, sk_none()) && !sk_less(tau, c)) {
        const uint32_t slot = atomicAdd(cnt, 1u);
        if (slot < kHybScratch) {
          sk[slot] = c.k;
          sx[slot] = c.i;
        }
      }
    }
    __syncthreads();
    const uint32_t S = *cnt;
    if (S <= kHybScratch) {
      if (threadIdx.x < S) {
        const SKey my{sk[threadIdx.x], sx[threadIdx.x]};
        uint32_t rank = 0;
        // (unrolled: eight LDS reads in flight -- one read per iteration is one LDS round trip per iteration)
#pragma unroll 8
        for (uint32_t o = 0; o < S; o++) rank += sk_less(SKey{sk[o], sx[o]}, my) ? 1u : 0u;
        if (rank < k) put(rank, my);
      }
    } else {
      direct = true;
    }
  }
  if (direct) {
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const SKey my{mk[j], mx[j]};
      if (!sk_same(my, sk_none())) {
        uint32_t rank = 0;
#pragma unroll 8
        for (uint32_t o = 0; o < n; o++) rank += sk_less(get(o), my) ? 1u : 0u;
        if (rank < k) put(rank, my);
      }
    }
  }
}

// The distances of a tile's nv compacted vector rows (vrow[j]: row number, in LDS) against the query staged in LDS (qs), their
// orderable keys to vkey[j].  Called by the whole workgroup of 256; Args: HybridTileArgs / HybridTreeArgs (the KNN fields).
template <int TYPE, int METRIC, typename Args>
__device__ __forceinline__ void hyb_knn_distances(const Args &A, const uint32_t *vrow, uint32_t *vkey, uint32_t nv, const u4 *qs) {
  const int G = A.G, ITERS = A.ITERS;
  const uint32_t gl = threadIdx.x & (uint32_t)(G - 1), grp = threadIdx.x / (uint32_t)G, GPB = 256u / (uint32_t)G;
  const u4 *__restrict__ rows = reinterpret_cast<const u4 *>(A.rows);
  // three rows per group and step, three chunks per row in flight (nine 16-byte loads per lane; four rows spill at six waves per SIMD): unconditional loads (from
  // chunk 0 where the lane has none, from the step's first row past the end), the operations of scan_kernel in its order --
  // chunk i of a lane is lane + i G, absent chunks are zeros, one Op::add per chunk slot i < ITERS, then the butterfly
  constexpr int RU = 3, CU = 3;
  if (ITERS <= CU && (A.knn_pipeline & 1)) {
    // Rows of at most CU chunks per lane (3 KiB fp32 rows at 64 lanes per row: every configs[4] row): ONE batch of loads per
    // step, so the NEXT step's rows are requested as soon as this step's have been multiplied in -- their round trip
    // (3.7 us under the kernel's own traffic) runs behind this step's butterfly, distance and store instead of after them.
    // Same loads, same Op::add order, same reduction tree: the gather's bits.
    u4 x[RU][CU];
    uint32_t cc[CU];
    bool ok[CU];
#pragma unroll
    for (int c = 0; c < CU; c++) {
      const uint32_t ch = gl + (uint32_t)c * (uint32_t)G;
      ok[c] = c < ITERS && ch < A.chunks;
      cc[c] = ok[c] ? ch : 0u;
    }
    auto request = [&](uint32_t j0) {
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const uint32_t ju = j0 + u * GPB;
        const u4 *pu = rows + (size_t)vrow[ju < nv ? ju : j0] * A.stride16;
#pragma unroll
        for (int c = 0; c < CU; c++) x[u][c] = load16<true>(pu + cc[c]);
      }
    };
    if (grp < nv) request(grp);
    for (uint32_t j0 = grp; j0 < nv; j0 += RU * GPB) {
      float acc[RU];
#pragma unroll
      for (int u = 0; u < RU; u++) acc[u] = 0.0f;
#pragma unroll
      for (int c = 0; c < CU; c++)
        if (c < ITERS) {
          const u4 q = ok[c] ? qs[cc[c]] : zero4();
#pragma unroll
          for (int u = 0; u < RU; u++) acc[u] = Op<TYPE, METRIC>::add(acc[u], ok[c] ? x[u][c] : zero4(), q);
        }
      // (unconditional -- past the end the step's first row again: a branch around the loads lets the scheduler sink them
      // to their use; the barriers pin them in front of the reduction they are meant to overlap)
      __builtin_amdgcn_sched_barrier(0);
      request(j0 + RU * GPB < nv ? j0 + RU * GPB : j0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const uint32_t ju = j0 + u * GPB;
        const float d = finish<TYPE, METRIC>(group_reduce_rt(acc[u], G), zero4());
        if (gl == 0 && ju < nv) vkey[ju] = f2key(d);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else
  for (uint32_t j0 = grp; j0 < nv; j0 += RU * GPB) {
    const u4 *p[RU];
#pragma unroll
    for (int u = 0; u < RU; u++) {
      const uint32_t ju = j0 + u * GPB;
      p[u] = rows + (size_t)vrow[ju < nv ? ju : j0] * A.stride16;
    }
    float acc[RU];
#pragma unroll
    for (int u = 0; u < RU; u++) acc[u] = 0.0f;
    for (int i0 = 0; i0 < ITERS; i0 += CU) {
      u4 x[RU][CU];
      uint32_t cc[CU];
      bool ok[CU];
#pragma unroll
      for (int c = 0; c < CU; c++) {
        const uint32_t ch = gl + (uint32_t)(i0 + c) * (uint32_t)G;
        ok[c] = i0 + c < ITERS && ch < A.chunks;
        cc[c] = ok[c] ? ch : 0u;
#pragma unroll
        for (int u = 0; u < RU; u++) x[u][c] = load16<true>(p[u] + cc[c]);  // (rows are read once: streamed past the caches)
      }
#pragma unroll
      for (int c = 0; c < CU; c++)
        if (i0 + c < ITERS) {
          const u4 q = ok[c] ? qs[cc[c]] : zero4();
#pragma unroll
          for (int u = 0; u < RU; u++) acc[u] = Op<TYPE, METRIC>::add(acc[u], ok[c] ? x[u][c] : zero4(), q);
        }
    }
#pragma unroll
    for (int u = 0; u < RU; u++) {
      const uint32_t ju = j0 + u * GPB;
      const float d = finish<TYPE, METRIC>(group_reduce_rt(acc[u], G), zero4());
      if (gl == 0 && ju < nv) vkey[ju] = f2key(d);
    }
  }
}

// One row's distance by its group of G lanes -- the gather's operations in its order (chunk i of a lane is lane + i G, absent
// chunks are zeros, one Op::add per chunk slot i < ITERS, then the butterfly): the bits of scan_kernel<GATHER>.
template <int TYPE, int METRIC>
__device__ __forceinline__ float hyb_row_distance(const void *rows_v, uint32_t stride16, uint32_t chunks, int G, int ITERS, uint32_t row,
                                                  const u4 *qs, uint32_t gl) {
  const u4 *__restrict__ p = reinterpret_cast<const u4 *>(rows_v) + (size_t)row * stride16;
  float acc = 0.0f;
  for (int i = 0; i < ITERS; i++) {
    const uint32_t ch = gl + (uint32_t)i * (uint32_t)G;
    const bool ok = ch < chunks;
    const u4 x = load16<true>(p + (ok ? ch : 0u));
    const u4 q = ok ? qs[ch] : zero4();
    acc = Op<TYPE, METRIC>::add(acc, ok ? x : zero4(), q);
  }
  return finish<TYPE, METRIC>(group_reduce_rt(acc, G), zero4());
}

// Multi-value indexes off identity labelling (A.L.next): a document's distance is the MINIMUM over its vectors
// (VecSimIndex_GetDistanceFrom_Unsafe of a multi index; FlatIndex::gather).  vkey[j] holds the key of the label's first row
// vrow[j]; the label's further rows follow through next[].  A NaN's key is the largest: a number beats it.
template <int TYPE, int METRIC, typename Args>
__device__ __forceinline__ void hyb_knn_chain_min(const Args &A, const uint32_t *vrow, uint32_t *vkey, uint32_t nv, const u4 *qs) {
  const int G = A.G;
  const uint32_t gl = threadIdx.x & (uint32_t)(G - 1), grp = threadIdx.x / (uint32_t)G, GPB = 256u / (uint32_t)G;
  for (uint32_t j = grp; j < nv; j += GPB) {
    uint32_t best = 0xFFFFFFFFu;
    uint32_t r = A.L.next[vrow[j]];
    for (uint32_t guard = 0; r < A.L.n_rows && guard < A.L.n_rows; guard++) {
      const float d = hyb_row_distance<TYPE, METRIC>(A.rows, A.stride16, A.chunks, G, A.ITERS, r, qs, gl);
      const uint32_t key = f2key(d);
      best = key < best ? key : best;
      r = A.L.next[r];
    }
    if (gl == 0 && best < vkey[j]) vkey[j] = best;
  }
}

// phase clock of a tile (knob hybrid_trace: where a workgroup's time goes; s_memrealtime ticks at 100 MHz)
#define RSGPU_HYB_MARK(p)                                                                                   \
  do {                                                                                                      \
    if (A.trace && threadIdx.x == 0) A.trace[(size_t)tile * kHybTracePhases + (p)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

// Dynamic LDS: pool_words u32 (the probe's window of another list; then the hits' records -- doc id | frequency in list 0 |
// position in list l -- then their keys | doc ids; then the vector rows | doc ids | distance keys of branch B), then the KNN
// query (chunks x 16 bytes).
// DPT drivers per thread (a tile = 256 DPT drivers), NL: the lists the instantiation holds positions for.  Only <kHybDpt,
// kHybMaxLists> is instantiated: round 5 measured tiles of 2 048 (one round over the chip instead of two: 63.7 us against 45.7 --
// every phase grows with the tile, the tiles that hold vectors take 55 us) and of 512 (54.2 us: the per-tile overhead) on the
// configs[4] stream, profiles/r05_hybrid_tile_size_ab.json.
// (the body of both kernels below: Args = HybridTileArgs, tile = blockIdx.x -- one query per grid -- or HybridTileLite and the
// tile the block maps to in a grid several queries share)
template <int TYPE, int METRIC, int DPT, int NL, typename Args>
__device__ __forceinline__ void hybrid_tile_body(const Args &A, const uint32_t tile) {
#include "hybrid_tile_body.inc"
}
#undef RSGPU_HYB_MARK
template <int TYPE, int METRIC, int DPT, int NL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void hybrid_tile_kernel(HybridTileArgs A) {
  const uint32_t tile = blockIdx.x;
#define RSGPU_HYB_MARK(p)                                                                                   \
  do {                                                                                                      \
    if (A.trace && threadIdx.x == 0) A.trace[(size_t)tile * kHybTracePhases + (p)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#include "hybrid_tile_body.inc"
#undef RSGPU_HYB_MARK
}
// Several queries in one grid (search_kernels.hpp HybridTileBatch).  Block -> (query, tile): the queries sit in ascending order of
// their tile counts; segment j = the rounds in which the queries j .. n_q - 1 still have tiles, its blocks deal those queries'
// next tiles out in turn -- every query's first tiles (the vector-bearing ones of a corpus whose vectors carry the low doc ids)
// start at once, behind them the light tiles of all queries fill the chip without a half-empty second round per query.
template <int TYPE, int METRIC, int DPT, int NL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void hybrid_tile_batch_kernel(HybridTileBatch B) {
  uint32_t j = 0;
#pragma unroll
  for (int i = 0; i + 1 < kHybBatchMax; i++)
    if (i + 1 < (int)B.n_q && blockIdx.x >= B.tile_end[i]) j = (uint32_t)i + 1;
  const uint32_t seg0 = j ? B.tile_end[j - 1] : 0u;
  uint32_t qi = j, tile = blockIdx.x - seg0;
  if (B.interleave) {
    const uint32_t alive = B.n_q - j, off = blockIdx.x - seg0;
    tile = (j ? B.n_tiles[j - 1] : 0u) + off / alive;
    qi = j + off % alive;
  }
  hybrid_tile_body<TYPE, METRIC, DPT, NL>(B.q[qi], tile);
}

// One workgroup of 1 024 per branch (block 0: the score lists; the last block: the hit count and the KNN lists), nothing shared
// between them.  The k best of the tiles' lists (a list is sorted: its first entry is the tile's best):
//   1. every thread keeps the best of ITS tiles' first entries (tile t belongs to thread t mod 1 024): 1 024 distinct entries;
//   2. every wavefront ranks its 64 and takes its k-th; the smallest of those 16 is the k-th of SOME k entries, so the k-th of
//      all is not above it; for k <= 16 the wavefronts' k best (16 k entries) are ranked against each other as well, which
//      gives the k-th of all 1 024 -- a bound that only about k tiles' firsts pass;
//   3. a winner's tile has its first entry at or below the bound: only THOSE tiles' lists are read (the first version walked
//      all tiles x k entries twice -- 600 KB through one CU), their entries at or below the bound are collected in LDS and
//      ranked; ranks below k are the answer.
// More survivors than the LDS list holds (an adversarial arrangement): *out_n = 0xFFFFFFFF and the caller answers the query
// with the staged pipeline.
constexpr uint32_t kHybSurvivors = 4096;  // (R.surv_cap <= this: a knob for the tests of the way out; 2 048 until round 6)
#define RSGPU_RED_MARK(p)                                                                                              \
  do {                                                                                                                 \
    if (R.trace && threadIdx.x == 0) R.trace[(SCORE ? 0 : kHybTracePhases) + (p)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
template <bool SCORE>
__device__ __forceinline__ void hybrid_reduce_branch(const HybridReduceArgs &R, uint64_t *lk, uint32_t *li, uint32_t *cnt_sh) {
  __shared__ uint64_t wtau_k[16];
  __shared__ uint32_t wtau_i[16];
  const uint32_t k = SCORE ? R.top_n : R.k;
  const uint32_t n_tiles = R.n_tiles;  // >= 1
  (void)0;
  // entry e -- the KNN composite carries its doc id in the low word, i is only the "none" mark there
  auto entry = [&](uint32_t e) {
    const uint64_t ck = SCORE ? R.part_skey[e] : R.part_knn[e];
    const uint32_t ci = ck == ~0ull ? ~0u : (SCORE ? R.part_sidx[e] : 0u);
    return SKey{ck, ci};
  };
  // the first entries of the thread's tiles t0 + j * 1024 (none past the end), kB loads in flight
  constexpr int kB = 8;
  auto load_firsts = [&](uint32_t t0, uint64_t (&ck)[kB], uint32_t (&ci)[kB]) {
#pragma unroll
    for (int j = 0; j < kB; j++) {
      const uint32_t t = t0 + j * 1024, tt = t < n_tiles ? t : n_tiles - 1;
      ck[j] = SCORE ? R.part_skey[(size_t)tt * k] : R.part_knn[(size_t)tt * k];
      ci[j] = SCORE ? R.part_sidx[(size_t)tt * k] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kB; j++) {
      if (t0 + j * 1024 >= n_tiles) ck[j] = ~0ull;
      if (ck[j] == ~0ull) ci[j] = ~0u;
    }
  };
  RSGPU_RED_MARK(0);
  // 1. the thread's best (the first kB of its tiles stay in registers for step 3)
  SKey best = sk_none();
  uint64_t ck0[kB];
  uint32_t ci0[kB];
  load_firsts(threadIdx.x, ck0, ci0);
#pragma unroll
  for (int j = 0; j < kB; j++) best = sk_min(SKey{ck0[j], ci0[j]}, best);
  for (uint32_t t0 = threadIdx.x + kB * 1024; t0 < n_tiles; t0 += kB * 1024) {
    uint64_t ck[kB];
    uint32_t ci[kB];
    load_firsts(t0, ck, ci);
#pragma unroll
    for (int j = 0; j < kB; j++) best = sk_min(SKey{ck[j], ci[j]}, best);
  }
  lk[threadIdx.x] = best.k;
  li[threadIdx.x] = best.i;
  if (threadIdx.x == 0) {
    *cnt_sh = 0;
    wtau_k[0] = ~0ull;
    wtau_i[0] = ~0u;
  }
  RSGPU_RED_MARK(1);  // firsts loaded
  __syncthreads();
  // 2. the bound: the k-th smallest of 64 GROUP minima (group g = the threads g, g + 64, ...: sixteen threads' bests, i.e. the
  // best first entry of a 64th of the tiles).  k groups have their minimum at or below it, so the k-th entry of all is not
  // above it; a tile that is not its group's best passes it with probability ~k / tiles, so about k tiles pass.  One
  // wavefront ranks 64 values -- the first version ranked the 1 024 thread bests inside every wavefront and 16 k of them
  // against each other: 9 of the kernel's 17 us on one CU (profiles/r04_hybrid_trace.txt).
  // (round 6: the 64 x 64 comparisons are shared among the sixteen wavefronts -- wavefront w compares every group minimum with the
  // four minima 4 w .. 4 w + 3, the shares add up in LDS; one wavefront alone counted 64 entries per lane, 2 of the branch's 11 us)
  __shared__ uint32_t grank[64];
  if (threadIdx.x < 64) {
    SKey g = sk_none();
#pragma unroll
    for (int j = 0; j < 16; j++) g = sk_min(SKey{lk[threadIdx.x + 64 * j], li[threadIdx.x + 64 * j]}, g);
    lk[1024 + threadIdx.x] = g.k;
    li[1024 + threadIdx.x] = g.i;
    grank[threadIdx.x] = 0;
  }
  __syncthreads();
  {
    const uint32_t l = threadIdx.x & 63, w4 = (threadIdx.x >> 6) * 4;
    const SKey g{lk[1024 + l], li[1024 + l]};
    if (!sk_same(g, sk_none())) {
      uint32_t rank = 0;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) rank += sk_less(SKey{lk[1024 + w4 + j], li[1024 + w4 + j]}, g) ? 1u : 0u;
      if (rank) atomicAdd(&grank[l], rank);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const SKey g{lk[1024 + threadIdx.x], li[1024 + threadIdx.x]};
    if (!sk_same(g, sk_none()) && grank[threadIdx.x] == k - 1) {
      wtau_k[0] = g.k;
      wtau_i[0] = g.i;
    }
  }
  __syncthreads();
  SKey tau{wtau_k[0], wtau_i[0]};
  if (sk_same(tau, sk_none())) {
    // fewer than k of the 64 groups hold anything (k above 32 over a few dozen tiles: round-5 advisor -- everything passed and
    // queries of ordinary size went to the exact-select settlement): the k-th of the 1 024 THREAD bests instead, every thread
    // counting the bests before its own -- rare, 1 024 LDS reads per thread.  Fewer than k tiles hold anything: no bound at all,
    // every entry is ranked (at most 63 x 64 of them: they fit the survivors' list)
    if (!sk_same(best, sk_none())) {
      uint32_t rank = 0;
#pragma unroll 8
      for (uint32_t j = 0; j < 1024; j++) rank += sk_less(SKey{lk[j], li[j]}, best) ? 1u : 0u;
      if (rank == k - 1) {
        wtau_k[0] = best.k;
        wtau_i[0] = best.i;
      }
    }
    __syncthreads();
    tau = SKey{wtau_k[0], wtau_i[0]};
  }
  __syncthreads();  // (lk / li are rewritten below)
  RSGPU_RED_MARK(2);  // bound
  // 3. the lists of the tiles whose first entry passes; survivors.  The passing tiles (about k of them) are listed first, then
  // ALL their entries are requested at once, one (tile, entry) pair per thread: walking a list entry by entry until one
  // fails the bound was up to k dependent memory round trips on the one workgroup the whole query waits for
  __shared__ uint32_t pass_t[1024];
  __shared__ uint32_t pass_n;
  if (threadIdx.x == 0) pass_n = 0;
  __syncthreads();
  for (uint32_t t0 = threadIdx.x; t0 < n_tiles; t0 += kB * 1024) {
    uint64_t ck[kB];
    uint32_t ci[kB];
    if (t0 == threadIdx.x) {
#pragma unroll
      for (int j = 0; j < kB; j++) {
        ck[j] = ck0[j];
        ci[j] = ci0[j];
      }
    } else {
      load_firsts(t0, ck, ci);
    }
#pragma unroll
    for (int j = 0; j < kB; j++) {
      const SKey first{ck[j], ci[j]};
      if (!sk_same(first, sk_none()) && !sk_less(tau, first)) {
        const uint32_t slot = atomicAdd(&pass_n, 1u);
        if (slot < 1024) pass_t[slot] = t0 + j * 1024;
      }
    }
  }
  __syncthreads();
  const uint32_t P = pass_n;
  RSGPU_RED_MARK(3);  // passing tiles listed
  if (P > 1024) {  // (more tiles at the bound than the list holds: the staged pipeline answers)
    if (threadIdx.x == 0) {
      *(SCORE ? R.out_sn : R.out_kn) = 0xFFFFFFFFu;
      __hip_atomic_store(R.done + (SCORE ? 0 : 1), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  for (uint32_t e0 = 0; e0 < P * k; e0 += 4 * 1024) {  // four loads in flight per thread
    SKey c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t e = e0 + j * 1024 + threadIdx.x;
      const uint32_t ec = e < P * k ? e : 0;
      c[j] = entry((uint32_t)((size_t)pass_t[ec / k] * k + ec % k));
      if (e >= P * k) c[j] = sk_none();
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (!sk_same(c[j], sk_none()) && !sk_less(tau, c[j])) {
        const uint32_t slot = atomicAdd(cnt_sh, 1u);
        if (slot < R.surv_cap) {
          lk[slot] = c[j].k;
          li[slot] = c[j].i;
        }
      }
  }
  __syncthreads();
  RSGPU_RED_MARK(4);  // their entries collected
  const uint32_t S = *cnt_sh;
  uint32_t *out_n = SCORE ? R.out_sn : R.out_kn;
  uint32_t *done = R.done + (SCORE ? 0 : 1);
  if (S > R.surv_cap) {
    if (threadIdx.x == 0) {
      *out_n = 0xFFFFFFFFu;
      __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  auto answer = [&](uint32_t rank, const SKey &my) {
    if (SCORE) {
      R.out_skeys[rank] = my.k;
      R.out_sids[rank] = my.i;
    } else {
      R.out_krows[rank] = (uint32_t)my.k;
      R.out_kkeys[rank] = (uint32_t)(my.k >> 32);
      R.out_kids[rank] = (uint32_t)my.k;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (the few threads that wrote an answer: it is in host memory before the flag)
  };
  if (S <= 512) {
    // the usual few dozen survivors: W columns (a power of two >= S) x P rows of the 1 024 threads, a row counts its share of the
    // survivors in front of each entry, the shares add up in LDS (round 6: the first S threads counted all S, the other wavefronts idle)
    __shared__ uint32_t srank[512];
    const uint32_t W = S <= 64 ? 64u : (S <= 128 ? 128u : (S <= 256 ? 256u : 512u)), P = 1024u / W;
    const uint32_t e = threadIdx.x & (W - 1), part = threadIdx.x / W, chunk = (S + P - 1) / P;
    const uint32_t lo = part * chunk, hi = lo + chunk < S ? lo + chunk : S;
    if (threadIdx.x < W) srank[threadIdx.x] = 0;
    __syncthreads();
    if (e < S && lo < hi) {
      const SKey my{lk[e], li[e]};
      uint32_t rank = 0;
#pragma unroll 4
      for (uint32_t j = lo; j < hi; j++) rank += sk_less(SKey{lk[j], li[j]}, my) ? 1u : 0u;
      if (rank) atomicAdd(&srank[e], rank);
    }
    __syncthreads();
    if (threadIdx.x < S && srank[threadIdx.x] < k) answer(srank[threadIdx.x], SKey{lk[threadIdx.x], li[threadIdx.x]});
  } else {
    for (uint32_t e = threadIdx.x; e < S; e += 1024) {
      const SKey my{lk[e], li[e]};
      uint32_t rank = 0;
#pragma unroll 8
      for (uint32_t j = 0; j < S; j++) rank += sk_less(SKey{lk[j], li[j]}, my) ? 1u : 0u;
      if (rank < k) answer(rank, my);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *out_n = S < k ? S : k;
    __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  RSGPU_RED_MARK(5);  // ranked, written
}
#undef RSGPU_RED_MARK

// role 0: the score lists, 1: the KNN lists, 2: the hit count
__device__ __forceinline__ void hybrid_reduce_body(const HybridReduceArgs &R, const uint32_t role) {
  __shared__ uint64_t lk[kHybSurvivors];
  __shared__ uint32_t li[kHybSurvivors];
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t cnt_sh;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (role == 2) {  // the hit count: a workgroup of its own (it was two memory round trips in front of the KNN branch)
    uint32_t s = 0;
    for (uint32_t t0 = threadIdx.x; t0 < R.n_tiles; t0 += 4 * 1024) {  // (four loads in flight)
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t t = t0 + j * 1024;
        v[j] = R.tile_hits[t < R.n_tiles ? t : R.n_tiles - 1];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) s += t0 + j * 1024 < R.n_tiles ? v[j] : 0u;
    }
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) wsum[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int j = 0; j < 16; j++) t += wsum[j];
      *R.out_hits = t;
      __hip_atomic_store(R.done + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (role == 0 && R.top_n) hybrid_reduce_branch<true>(R, lk, li, &cnt_sh);
  else if (role == 1 && R.k) hybrid_reduce_branch<false>(R, lk, li, &cnt_sh);
}
__global__ __launch_bounds__(1024) void hybrid_reduce_kernel(HybridReduceArgs R) {
  // (the grid holds the enabled branches only: block 0 is the score branch when there is one, the last block the hit count)
  hybrid_reduce_body(R, blockIdx.x == gridDim.x - 1 ? 2u : (blockIdx.x == 0 && R.top_n ? 0u : 1u));
}
// the reduce launches of the queries of a shared grid in one: three workgroups per query (a disabled branch's returns at once)
__global__ __launch_bounds__(1024) void hybrid_reduce_batch_kernel(HybridReduceBatch B) {
  hybrid_reduce_body(B.q[blockIdx.x / 3u], blockIdx.x % 3u);
}



// ---- the general form: a root intersection over terms / unions / intersections of terms, max_slop / in_order, per-hit slop, the
// ordered hit list (search_kernels.hpp HybridTreeArgs) ----------------------------------------------------------------------------
constexpr uint32_t kHybNone = 0xFFFFFFFFu;

// Ordered compaction of the workgroup's flagged slots: slot (k, thread) -- k-major: the order of the drivers, and of compacted
// entries k * 256 + thread -- gets the number of flagged slots before it; returns their total.  seg: DPT * 4 + 1 words of LDS,
// free again once every thread has passed a later barrier.
template <int DPT>
__device__ __forceinline__ uint32_t ordered_slots(const bool (&flag)[DPT], uint32_t (&slot)[DPT], uint32_t *seg) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long m[DPT];
#pragma unroll
  for (int k = 0; k < DPT; k++) {
    m[k] = __ballot(flag[k]);
    if (lane == 0) seg[k * 4 + wave] = (uint32_t)__popcll(m[k]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int q = 0; q < DPT * 4; q++) {
      const uint32_t c = seg[q];
      seg[q] = run;
      run += c;
    }
    seg[DPT * 4] = run;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < DPT; k++) slot[k] = seg[k * 4 + wave] + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull));
  return seg[DPT * 4];
}

// Dynamic LDS as hybrid_tile_kernel: pool_words u32 -- a probed list's window; then the candidates' records: doc id |
// entry index per LEAF ((n + 1) arrays of a tile's drivers), the frequencies in the entry indices' place once they are
// gathered; then keys | doc ids; then branch B's rows | doc ids | keys -- and the KNN query behind it.
// ML: the lists (leaves + excluded) the instantiation holds positions / cursors for -- 4 (79 VGPRs: six wavefronts per SIMD) for
// the queries of up to four lists that read term offsets, kHybTreeMaxLists = 8 (85: five) for the rest of those; without the
// proximity cursors (PROX = false) eight lists fit 74-84 registers and one instantiation serves every query
// DEEP: the result tree has more levels than root -> children -> terms (ScoreParams::n_nodes > 0, at most kHybDeepLevels below the
// root): the score folds it node by node (score_one<true>), accumulators in registers
// PROX: the query reads term offsets (a window to check, a scorer that divides by the slop): only these instantiations carry the
// proximity cursors (ProxCtx: scratch memory); the others are scratch-free
template <int TYPE, int METRIC, int ML, bool DEEP = false, bool PROX = true>
__global__ __launch_bounds__(256) void hybrid_tree_tile_kernel(HybridTreeArgs A) {
  constexpr int DPT = kHybDpt;
  constexpr uint32_t TILE = kHybTile;
  extern __shared__ __attribute__((aligned(16))) uint32_t win[];
  __shared__ uint32_t seg[DPT * 4 + 1];
  __shared__ uint32_t w_lo, w_hi, nv_sh, sel_cnt;
  __shared__ uint64_t sel_k[kHybScratch], sel_wtk[4];
  __shared__ uint32_t sel_x[kHybScratch], sel_wtx[4];
  const uint32_t WIN = A.pool_words;
  u4 *qs = reinterpret_cast<u4 *>(win + WIN);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n0 = A.len[0];
  const uint32_t i_first = blockIdx.x * TILE, i_next = i_first + TILE;
  const uint32_t *__restrict__ ids0 = A.ids[0];
  if (threadIdx.x == 0) nv_sh = 0;
  if (A.k)
    for (uint32_t c = threadIdx.x; c < A.chunks; c += 256) qs[c] = reinterpret_cast<const u4 *>(A.query)[c];

  bool live[DPT];
  uint32_t xc[DPT];           // doc id in the frame the lists share
  uint32_t ps[DPT][ML - 1];   // match position in list l, kHybNone: list l does not hold the document
  uint32_t mb[DPT];           // bit l: list l holds it
#pragma unroll
  for (int k = 0; k < DPT; k++) {
    const uint32_t i = i_first + k * 256 + threadIdx.x;
    live[k] = i < n0;
    const uint32_t ic = live[k] ? i : n0 - 1;
    xc[k] = (uint32_t)((long long)ids0[ic] + A.add[0]);
    mb[k] = live[k] ? 1u : 0u;
#pragma unroll
    for (int l = 0; l < ML - 1; l++) ps[k][l] = kHybNone;
  }
  const uint32_t x_first = (uint32_t)((long long)ids0[i_first] + A.add[0]);
  const uint32_t x_next = (uint32_t)((long long)ids0[i_next < n0 ? i_next : n0 - 1] + A.add[0]);

  // ---- probe: every other list through its window (hybrid_tile_kernel's, without the early exit: a union's lists are
  // alternatives, a driver that misses one may still be a candidate) ----
#pragma unroll
  for (int l = 1; l < ML; l++) {
    if (l < A.n) {
      const uint32_t *__restrict__ a = A.ids[l];
      const uint32_t nl = A.len[l];
      const long long add = A.add[l];
      if (A.dir[l]) {
        if (threadIdx.x == 0) {
          bool u0, u1;
          const uint32_t xf = to_list_frame(x_first, add, &u0), xn = to_list_frame(x_next, add, &u1);
          const uint32_t sh = A.dir_shift[l], dn = A.dir_n[l];
          const uint32_t bf = xf >> sh, bn = (uint32_t)min((uint64_t)(xn >> sh) + 1ull, (uint64_t)dn - 1ull);  // (xn = 2^32 - 1 at shift 0 must not wrap)
          const uint32_t dlo = A.dir[l][bf < dn ? bf : dn - 1], dhi = A.dir[l][bn < dn ? bn : dn - 1];
          w_lo = dlo;
          w_hi = i_next < n0 ? dhi : nl;
        }
      } else if (wave == 0) {
        bool u0;
        uint32_t rlo, rhi;
        wave_lower_bound_range(a, nl, to_list_frame(x_first, add, &u0), lane, 0u, false, &rlo, &rhi);
        if (lane == 0) w_lo = rlo;
      } else if (wave == 1) {
        bool u1;
        uint32_t rlo, rhi = nl;
        if (i_next < n0) wave_lower_bound_range(a, nl, to_list_frame(x_next, add, &u1), lane, 0u, false, &rlo, &rhi);
        if (lane == 0) w_hi = rhi;
      }
      __syncthreads();
      const uint32_t lo = w_lo, hi = w_hi;
      const uint32_t span = hi - lo;
      if (span <= WIN) {
        for (uint32_t base = 0; base < span; base += 8 * 256) {
          uint32_t t[8];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const uint32_t o = base + j * 256 + threadIdx.x;
            t[j] = a[lo + (o < span ? o : span - 1)];
          }
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const uint32_t o = base + j * 256 + threadIdx.x;
            if (o < span) win[o] = t[j];
          }
        }
        __syncthreads();
        uint32_t xl[DPT], b[DPT];
        bool under[DPT];
#pragma unroll
        for (int k = 0; k < DPT; k++) {
          xl[k] = to_list_frame(xc[k], add, &under[k]);
          b[k] = 0;
        }
        uint32_t rem = span;
        while (rem > 1) {
          const uint32_t half = rem >> 1;
#pragma unroll
          for (int k = 0; k < DPT; k++) b[k] = win[b[k] + half - 1] < xl[k] ? b[k] + half : b[k];
          rem -= half;
        }
#pragma unroll
        for (int k = 0; k < DPT; k++) {
          if (span && win[b[k]] < xl[k]) b[k]++;
          const bool m = live[k] && !under[k] && b[k] < span && win[b[k] < span ? b[k] : 0] == xl[k];
          if (m) {
            ps[k][l - 1] = lo + b[k];
            mb[k] |= 1u << l;
          }
        }
      } else {  // a window that does not fit: a binary search in memory, confined to the window
#pragma unroll
        for (int k = 0; k < DPT; k++) {
          bool under;
          const uint32_t x = to_list_frame(xc[k], add, &under);
          uint32_t b = lo, e = hi;
          if (live[k]) {
            while (b < e) {
              const uint32_t mid = b + ((e - b) >> 1);
              if (a[mid] < x) b = mid + 1;
              else e = mid;
            }
            if (!under && b < nl && a[b] == x) {
              ps[k][l - 1] = b;
              mb[k] |= 1u << l;
            }
          }
        }
      }
      __syncthreads();  // win / w_lo / w_hi are reused
    }
  }

  // ---- candidates: a list of every required set holds the document; compacted in driver order ----
  bool hit[DPT];
  if (DEEP && A.tree_pred) {
    // the match folded over the result tree (HybridTreeArgs::tree_pred): bottom-up which nodes match -- per level the AND / OR of
    // the children seen so far, as score_one<DEEP> keeps its sums --, then top-down which of them are in the result
    uint32_t lf[DPT], all_[DPT], any_[DPT];
    uint64_t matched[DPT];
#pragma unroll
    for (int k = 0; k < DPT; k++) {
      lf[k] = 0u;
#pragma unroll
      for (int l = 0; l < ML; l++)
        if (l < A.n_leaves) lf[k] |= ((mb[k] >> l) & 1u) << A.leaf_of[l];
      all_[k] = ~0u;
      any_[k] = 0u;
      matched[k] = 0ull;
    }
    const int nn = A.P.n_nodes;
    for (int i = 0; i < nn; i++) {
      const uint32_t d = A.P.node_depth[i], op = A.P.node_op[i], leaf = A.P.node_leaf[i];
#pragma unroll
      for (int k = 0; k < DPT; k++) {
        uint32_t v;
        if (op == 0u) {
          v = (lf[k] >> leaf) & 1u;
        } else {
          v = ((op == 2u ? all_[k] : any_[k]) >> d) & 1u;
          all_[k] |= 1u << d;
          any_[k] &= ~(1u << d);
        }
        if (d) {
          all_[k] &= ~((v ^ 1u) << (d - 1u));
          any_[k] |= v << (d - 1u);
        }
        matched[k] |= (uint64_t)v << i;
      }
    }
    uint32_t alive[DPT], present[DPT];
#pragma unroll
    for (int k = 0; k < DPT; k++) alive[k] = present[k] = 0u;
    for (int i = nn - 1; i >= 0; i--) {
      const uint32_t d = A.P.node_depth[i], op = A.P.node_op[i], leaf = A.P.node_leaf[i];
#pragma unroll
      for (int k = 0; k < DPT; k++) {
        const uint32_t up = d ? (alive[k] >> (d - 1u)) & 1u : 1u;
        const uint32_t in = (uint32_t)(matched[k] >> i) & 1u & up;
        alive[k] = (alive[k] & ~(1u << d)) | (in << d);
        if (op == 0u) present[k] |= in << leaf;
      }
    }
#pragma unroll
    for (int k = 0; k < DPT; k++) {
      hit[k] = live[k] && nn > 0 && ((matched[k] >> (nn - 1)) & 1ull) != 0ull && (mb[k] & A.veto) == 0u;
#pragma unroll
      for (int l = 1; l < ML; l++)
        if (l < A.n_leaves && !((present[k] >> A.leaf_of[l]) & 1u)) ps[k][l - 1] = kHybNone;
    }
  } else {
#pragma unroll
  for (int k = 0; k < DPT; k++) {
    // a nested / later child intersection that does not match as a whole is not in the result: its terms are absent
    uint32_t gone = 0u;
#pragma unroll
    for (int r = 0; r < ML; r++)
      if (r < A.n_opt_all && (mb[k] & A.opt_all[r]) != A.opt_all[r]) gone |= A.opt_all[r];
    const uint32_t held = mb[k] & ~gone;
    bool h = live[k];
#pragma unroll
    for (int r = 0; r < ML; r++)
      if (r < A.n_req) h = h && (held & A.req[r]) != 0u;
    h = h && (mb[k] & A.veto) == 0u;  // (NOT children: not.rs:171-209 -- the document must be absent from every excluded list)
#pragma unroll
    for (int r = 0; r < ML; r++)
      if (r < A.n_veto_all) h = h && (mb[k] & A.veto_all[r]) != A.veto_all[r];  // (root union: an earlier child's pass reports it)
    hit[k] = h;
#pragma unroll
    for (int l = 1; l < ML; l++)
      if ((gone >> l) & 1u) ps[k][l - 1] = kHybNone;
  }
  }
  uint32_t slot[DPT];
  const uint32_t nc = ordered_slots<DPT>(hit, slot, seg);
#pragma unroll
  for (int k = 0; k < DPT; k++)
    if (hit[k]) {
      const uint32_t s = slot[k];
      win[s] = xc[k];
      win[(1u + A.leaf_of[0]) * TILE + s] = i_first + k * 256 + threadIdx.x;
#pragma unroll
      for (int l = 1; l < ML; l++)
        if (l < A.n && A.leaf_of[l] != 0xFFu) win[(1u + A.leaf_of[l]) * TILE + s] = ps[k][l - 1];
    }
  __syncthreads();

  // ---- max_slop / in_order (Intersection::current_is_relevant, intersection.rs:205-215): prox_filter_kernel's test, one
  // candidate per lane where they sit compacted; the survivors close ranks, still in driver order ----
  uint32_t nh = nc;
  if (PROX && A.prox_filter) {
    bool keep[DPT];
    uint32_t ex_[DPT], ee[DPT][ML];
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const uint32_t e = j * 256 + threadIdx.x;
      keep[j] = false;
      ex_[j] = 0;
#pragma unroll
      for (int t = 0; t < ML; t++) ee[j][t] = kHybNone;
      if (e < nc) {
        ex_[j] = win[e];
#pragma unroll
        for (int t = 0; t < ML; t++)
          if (t < A.n_leaves) ee[j][t] = win[(1u + t) * TILE + e];
        if (prox_two_terms(A.X)) {  // (two plain terms: the cursors in registers, postings_ops.hpp)
          keep[j] = prox_within_range2(A.X, prox_term(A.O, 0, win[1u * TILE + e]), prox_term(A.O, 1, win[2u * TILE + e]));
        } else {
          ProxCtx<ML> x;
          prox_load<ML>(A.X, A.O, x, [&](int t) { return win[(1u + (uint32_t)t) * TILE + e]; });
          keep[j] = prox_within_range<ML>(A.X, x);
        }
      }
    }
    nh = ordered_slots<DPT>(keep, slot, seg);  // (its barriers: every record has been read before one is rewritten)
#pragma unroll
    for (int j = 0; j < DPT; j++)
      if (keep[j]) {
        const uint32_t s = slot[j];
        win[s] = ex_[j];
#pragma unroll
        for (int t = 0; t < ML; t++)
          if (t < A.n_leaves) win[(1u + t) * TILE + s] = ee[j][t];
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) A.tile_hits[blockIdx.x] = nh;

  // ---- per hit: slop, frequencies (per leaf, 0 where a union's term is absent), the hit list's records, the score ----
  uint64_t my_k[DPT];
  uint32_t my_x[DPT];
  const size_t g0 = (size_t)blockIdx.x * TILE;
#pragma unroll
  for (int j = 0; j < DPT; j++) {
    const uint32_t e = j * 256 + threadIdx.x;
    my_k[j] = ~0ull;
    my_x[j] = ~0u;
    if (e < nh) {
      const uint32_t x = win[e];
      my_x[j] = x;
      int slop = A.P.slop;
      if (PROX && A.prox_slop && A.top_n) {  // IndexResult_MinOffsetDelta from the term offsets (prox_slop_kernel's)
        if (prox_two_terms(A.X) && !A.X.count_present) {
          slop = prox_min_offset_delta2(prox_term(A.O, 0, win[1u * TILE + e]), prox_term(A.O, 1, win[2u * TILE + e]));
        } else {
          ProxCtx<ML> c;
          prox_load<ML>(A.X, A.O, c, [&](int t) { return win[(1u + (uint32_t)t) * TILE + e]; });
          slop = prox_min_offset_delta<ML>(A.X, c);
        }
      }
#pragma unroll
      for (int t = 0; t < ML; t++)
        if (t < A.n_leaves) {
          const uint32_t ep = win[(1u + t) * TILE + e];
          const uint32_t *__restrict__ fq = A.lfreq[t];
          // (a codec that stores no frequency yields the term record's default, 1: intersect_write_kernel)
          const uint32_t fv = ep == kHybNone ? 0u : (fq ? fq[ep] : 1u);
          if (A.hit_ids) {
            A.hit_freqs[(size_t)t * A.hit_stride + g0 + e] = fv;
            if (A.hit_epos) A.hit_epos[(size_t)t * A.hit_stride + g0 + e] = ep;
          }
          win[(1u + t) * TILE + e] = fv;  // (this lane's own record: nobody else reads it)
        }
      if (A.hit_ids) A.hit_ids[g0 + e] = x;
      if (A.top_n) {
        const long long tid = (long long)x + A.P.table_off;
        const bool known = tid >= 0 && tid < (long long)A.table_n;
        const uint32_t id = known ? (uint32_t)tid : 0u;
        float dscore;
        uint32_t dlen;
        if (A.len_score) {
          const uint2 ls = A.len_score[id];
          dlen = known ? ls.x : 0u;
          dscore = known ? __uint_as_float(ls.y) : 0.0f;
        } else {
          dscore = known ? A.doc_score[id] : 0.0f;
          dlen = known ? A.doc_len[id] : 0u;
        }
        const uint32_t mfreq = (known && A.max_freq) ? A.max_freq[id] : 0u;
        auto F = [&](int t) { return (double)win[(1u + (uint32_t)t) * TILE + e]; };
        const double s = score_one<DEEP, 0, DEEP ? kHybDeepLevels : kMaxTreeDepth>(A.P, F, dlen, dscore, mfreq, slop);
        my_k[j] = ~d2key(s);
      }
    }
  }
  __syncthreads();  // every record has been read: keys | doc ids (branch A), rows | doc ids | keys (branch B) take their place

  // ---- branch A: the tile's top-N ----
  if (A.top_n) {
    uint64_t *ek = reinterpret_cast<uint64_t *>(win);
    uint32_t *ex = win + 2 * TILE;
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const uint32_t e = j * 256 + threadIdx.x;
      if (e < nh) {
        ek[e] = my_k[j];
        ex[e] = my_x[j];
      }
    }
    __syncthreads();
    uint32_t sx_[DPT];  // (tile_select takes "none" past the end in both components)
#pragma unroll
    for (int j = 0; j < DPT; j++) sx_[j] = my_k[j] == ~0ull ? ~0u : my_x[j];
    tile_select<DPT>(my_k, sx_, nh, A.top_n, [&](uint32_t o) { return SKey{ek[o], ex[o]}; }, sel_k, sel_x, &sel_cnt, sel_wtk, sel_wtx,
                     [&](uint32_t rank, const SKey &my) {
                       A.part_skey[(size_t)blockIdx.x * A.top_n + rank] = my.k;
                       A.part_sidx[(size_t)blockIdx.x * A.top_n + rank] = my.i;
                     }, !PROX && (A.knn_pipeline & 2) != 0);  // (the instantiations that carry the proximity cursors sit at their register budget: 80)
    if (threadIdx.x >= nh && threadIdx.x < A.top_n) {
      A.part_skey[(size_t)blockIdx.x * A.top_n + threadIdx.x] = ~0ull;
      A.part_sidx[(size_t)blockIdx.x * A.top_n + threadIdx.x] = ~0u;
    }
    __syncthreads();
  }

  // ---- branch B: the hits that have a vector, their distances, the tile's top-k ----
  if (A.k) {
    uint32_t *vrow = win, *vx = win + TILE, *vkey = win + 2 * TILE;
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const uint32_t e = j * 256 + threadIdx.x;
      const uint32_t vr = e < nh ? label_first_row(A.L, A.ids_base + my_x[j]) : kNoRow;
      const bool has = vr != kNoRow;
      const unsigned long long m = __ballot(has);
      if (m) {
        uint32_t first = 0;
        const int leader = __builtin_ctzll(m);
        if (lane == (uint32_t)leader) first = atomicAdd(&nv_sh, (uint32_t)__popcll(m));
        first = __shfl(first, leader, 64);
        if (has) {
          const uint32_t s = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          vrow[s] = vr;
          vx[s] = my_x[j];
        }
      }
    }
    __syncthreads();
    const uint32_t nv = nv_sh;
    hyb_knn_distances<TYPE, METRIC>(A, vrow, vkey, nv, qs);
    __syncthreads();
    if (A.L.next) {
      hyb_knn_chain_min<TYPE, METRIC>(A, vrow, vkey, nv, qs);
      __syncthreads();
    }
    uint64_t vk_mine[DPT];
    uint32_t vx_mine[DPT];
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const uint32_t e = j * 256 + threadIdx.x;
      vk_mine[j] = e < nv ? (uint64_t)vkey[e] : ~0ull;
      vx_mine[j] = e < nv ? vx[e] : ~0u;
    }
    tile_select<DPT>(vk_mine, vx_mine, nv, A.k, [&](uint32_t o) { return SKey{(uint64_t)vkey[o], vx[o]}; }, sel_k, sel_x, &sel_cnt,
                     sel_wtk, sel_wtx, [&](uint32_t rank, const SKey &my) {
                       A.part_knn[(size_t)blockIdx.x * A.k + rank] = (my.k << 32) | my.i;
                     }, !PROX && (A.knn_pipeline & 2) != 0);  // (the instantiations that carry the proximity cursors sit at their register budget: 80)
    if (threadIdx.x >= nv && threadIdx.x < A.k) A.part_knn[(size_t)blockIdx.x * A.k + threadIdx.x] = ~0ull;
  }
}

// The staged pipeline's form of hyb_knn_chain_min: dists[i] (of the label's first row first_rows[i], written by the gather)
// becomes the minimum over the label's rows.  The query is staged in LDS; one group of G lanes per candidate.
struct KnnChainArgs {
  const void *rows;
  uint32_t stride16, chunks;
  int G, ITERS;
  const void *query;
  const uint32_t *first_rows;
  uint32_t m;
  const uint32_t *m_dev;
  LabelRows L;
  float *dists;
};
template <int TYPE, int METRIC>
__global__ __launch_bounds__(256) void knn_chain_min_kernel(KnnChainArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_q[];
  u4 *qs = reinterpret_cast<u4 *>(lds_q);
  for (uint32_t c = threadIdx.x; c < A.chunks; c += 256) qs[c] = reinterpret_cast<const u4 *>(A.query)[c];
  __syncthreads();
  const int G = A.G;
  const uint32_t gl = threadIdx.x & (uint32_t)(G - 1), grp = threadIdx.x / (uint32_t)G, GPB = 256u / (uint32_t)G;
  uint32_t m = A.m;
  if (A.m_dev) {
    const uint32_t md = *A.m_dev;
    m = md < m ? md : m;
  }
  for (uint32_t i = blockIdx.x * GPB + grp; i < m; i += gridDim.x * GPB) {
    const uint32_t r0 = A.first_rows[i];
    if (r0 >= A.L.n_rows) continue;
    float best = A.dists[i];
    bool changed = false;
    uint32_t r = A.L.next[r0];
    for (uint32_t guard = 0; r < A.L.n_rows && guard < A.L.n_rows; guard++) {
      const float d = hyb_row_distance<TYPE, METRIC>(A.rows, A.stride16, A.chunks, G, A.ITERS, r, qs, gl);
      if (best != best || d < best) {  // (FlatIndex::gather: a number beats a NaN)
        best = d;
        changed = true;
      }
      r = A.L.next[r];
    }
    if (gl == 0 && changed) A.dists[i] = best;
  }
}

// Settling a mass tie across the passes of one query (search_abi.cpp hyb_settle_overflow): the doc ids of the entries whose score
// key is tau, "none" for every other entry -- the exact select then takes the smallest of them.
__global__ __launch_bounds__(256) void hybrid_tie_ids_kernel(const uint64_t *__restrict__ skey, const uint32_t *__restrict__ sidx, uint32_t n,
                                                             uint64_t tau, uint64_t *__restrict__ out) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e < n) out[e] = skey[e] == tau ? (uint64_t)sidx[e] : ~0ull;
}

// The hit list out of the tiles' fixed slots: tile t's hits go behind those of the tiles before it.  One workgroup per tile; it
// sums the counts below its own (n_tiles <= 16 Ki words, L2-resident).  Runs BEHIND the reduce kernel: the query's answers are
// in host memory before this kernel starts.
__global__ __launch_bounds__(256) void hybrid_hits_pack_kernel(const uint32_t *__restrict__ tile_hits, uint32_t n_tiles, int n_leaves,
                                                               const uint32_t *__restrict__ src_ids, const uint32_t *__restrict__ src_freqs,
                                                               const uint32_t *__restrict__ src_epos, uint32_t src_stride,
                                                               uint32_t *__restrict__ dst_ids, uint32_t *__restrict__ dst_freqs,
                                                               uint32_t *__restrict__ dst_epos, uint32_t dst_cap, uint32_t *total_out,
                                                               HybridRuns runs) {
  __shared__ uint32_t wsum[4];
  const uint32_t t = blockIdx.x;
  uint32_t s = 0;
  for (uint32_t u = threadIdx.x; u < t; u += 256) s += tile_hits[u];
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  const uint32_t off = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const uint32_t cnt = tile_hits[t];
  const size_t from = (size_t)t * kHybTile;
  for (uint32_t r = threadIdx.x; r < cnt; r += 256) {
    dst_ids[off + r] = src_ids[from + r];
    for (int l = 0; l < n_leaves; l++) {
      dst_freqs[(size_t)l * dst_cap + off + r] = src_freqs[(size_t)l * src_stride + from + r];
      if (src_epos) dst_epos[(size_t)l * dst_cap + off + r] = src_epos[(size_t)l * src_stride + from + r];
    }
  }
  if (t == n_tiles - 1 && threadIdx.x == 0) *total_out = off + cnt;
  if (runs.n && threadIdx.x == 0) {
    for (uint32_t p = 0; p < runs.n; p++)
      if (runs.first_tile[p] == t) runs.run_start[p] = off;
    if (t == n_tiles - 1) runs.run_start[runs.n] = off + cnt;
  }
}

// (search_kernels.hpp launch_hybrid_hits_merge)
__global__ __launch_bounds__(256) void hybrid_hits_merge_kernel(HybridRuns runs, int n_leaves, const uint32_t *__restrict__ src_ids,
                                                                const uint32_t *__restrict__ src_freqs, const uint32_t *__restrict__ src_epos,
                                                                uint32_t src_cap, uint32_t *__restrict__ dst_ids, uint32_t *__restrict__ dst_freqs,
                                                                uint32_t *__restrict__ dst_epos, uint32_t dst_cap) {
  uint32_t start[9];
#pragma unroll
  for (int p = 0; p < 9; p++) start[p] = p <= (int)runs.n ? runs.run_start[p] : 0u;
  const uint32_t total = runs.run_start[runs.n];
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256) {
    const uint32_t x = src_ids[j];
    uint32_t dest = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (q < (int)runs.n) {
        const uint32_t b0 = start[q], e0 = start[q + 1];
        if (j >= b0 && j < e0) {
          dest += j - b0;
        } else {
          uint32_t b = b0, e = e0;
          while (b < e) {
            const uint32_t mid = b + ((e - b) >> 1);
            if (src_ids[mid] < x) b = mid + 1;
            else e = mid;
          }
          dest += b - b0;
        }
      }
    }
    dst_ids[dest] = x;
    for (int l = 0; l < n_leaves; l++) {
      dst_freqs[(size_t)l * dst_cap + dest] = src_freqs[(size_t)l * src_cap + j];
      if (src_epos) dst_epos[(size_t)l * dst_cap + dest] = src_epos[(size_t)l * src_cap + j];
    }
  }
}

// dir[b] = lower_bound(ids, b << shift): entry i owns the buckets behind its predecessor's up to its own (the first entry
// the buckets from 0, the last one also those behind its own up to dir_n - 1, which hold n)
__global__ __launch_bounds__(256) void build_bucket_dir_kernel(const uint32_t *__restrict__ ids, uint32_t n, uint32_t shift,
                                                               uint32_t *__restrict__ dir, uint32_t dir_n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t b = ids[i] >> shift;
    uint32_t from = i ? (ids[i - 1] >> shift) + 1 : 0;
    for (; from <= b && from < dir_n; from++) dir[from] = i;
    if (i + 1 == n)
      for (uint32_t t = b + 1; t < dir_n; t++) dir[t] = n;
  }
}
__global__ __launch_bounds__(256) void pack_len_score_kernel(const uint32_t *__restrict__ doc_len, const float *__restrict__ doc_score,
                                                             uint32_t n, uint2 *__restrict__ ls) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) ls[i] = make_uint2(doc_len[i], __float_as_uint(doc_score[i]));
}

}  // namespace

void launch_build_bucket_dir(const uint32_t *ids, uint32_t n, uint32_t shift, uint32_t *dir, uint32_t dir_n, hipStream_t s) {
  if (!n || !dir_n) return;
  const uint32_t need = (n + 255) / 256;
  hipLaunchKernelGGL(build_bucket_dir_kernel, dim3(need < 4096 ? need : 4096), dim3(256), 0, s, ids, n, shift, dir, dir_n);
}
void launch_pack_len_score(const uint32_t *doc_len, const float *doc_score, uint32_t n, void *ls, hipStream_t s) {
  if (!n) return;
  const uint32_t need = (n + 255) / 256;
  hipLaunchKernelGGL(pack_len_score_kernel, dim3(need < 8192 ? need : 8192), dim3(256), 0, s, doc_len, doc_score, n, (uint2 *)ls);
}

uint32_t hybrid_tiles(uint32_t n0) { return (n0 + kHybTile - 1) / kHybTile; }
bool hybrid_tile_supported(int type, int metric, uint32_t stride16, uint32_t n_tiles, uint32_t top_n, uint32_t k) {
  if (top_n > (uint32_t)kHybMaxK || k > (uint32_t)kHybMaxK) return false;
  if (n_tiles > kHybMaxTiles) return false;  // (the reduce workgroup walks tiles x k entries: 16 Ki tiles = 16 M drivers)
  if (!k) return true;
  if (type != KT_F32 && type != KT_F16 && type != KT_BF16) return false;
  if (metric != KM_L2 && metric != KM_IP) return false;
  return stride16 >= 1 && stride16 <= (uint32_t)kHybMaxChunks;
}

void launch_hybrid_tiles(const HybridTileArgs &args, int type, int metric, uint32_t n_tiles, hipStream_t s) {
  if (!n_tiles) return;
  HybridTileArgs a = args;
  if (a.k) {  // the lanes-per-row shape the scan kernels give this row length
    const Shape sh = pick_shape(a.stride16);
    a.G = sh.G;
    a.ITERS = sh.ITERS;
  } else {
    a.G = 1;
    a.ITERS = 0;
  }
  // LDS: the pool -- a window of 4 Ki entries; (lists + 1) record arrays of a tile's hits -- and the KNN query
  a.pool_words = std::max<uint32_t>(4096u, (uint32_t)(a.n + 1) * kHybTile);
  const size_t lds = (size_t)a.pool_words * 4 + (a.k ? (size_t)a.chunks * 16 : 0);
#define RSGPU_HYB(T, M) hipLaunchKernelGGL((hybrid_tile_kernel<T, M, kHybDpt, kHybMaxLists>), dim3(n_tiles), dim3(256), lds, s, a)
  if (!a.k) RSGPU_HYB(KT_F32, KM_IP);  // (no KNN branch: any instantiation)
  else if (type == KT_F32 && metric == KM_L2) RSGPU_HYB(KT_F32, KM_L2);
  else if (type == KT_F32) RSGPU_HYB(KT_F32, KM_IP);
  else if (type == KT_F16 && metric == KM_L2) RSGPU_HYB(KT_F16, KM_L2);
  else if (type == KT_F16) RSGPU_HYB(KT_F16, KM_IP);
  else if (metric == KM_L2) RSGPU_HYB(KT_BF16, KM_L2);
  else RSGPU_HYB(KT_BF16, KM_IP);
#undef RSGPU_HYB
}
bool hybrid_tree_supported(int type, int metric, uint32_t stride16, uint32_t n_tiles, uint32_t top_n, uint32_t k, int n_lists) {
  return n_lists >= 1 && n_lists <= kHybTreeMaxLists && hybrid_tile_supported(type, metric, stride16, n_tiles, top_n, k);
}
void launch_hybrid_tree_tiles(const HybridTreeArgs &args, int type, int metric, uint32_t n_tiles, hipStream_t s) {
  if (!n_tiles) return;
  HybridTreeArgs a = args;
  if (a.k) {
    const Shape sh = pick_shape(a.stride16);
    a.G = sh.G;
    a.ITERS = sh.ITERS;
  } else {
    a.G = 1;
    a.ITERS = 0;
  }
  a.pool_words = std::max<uint32_t>(4096u, (uint32_t)(a.n_leaves + 1) * kHybTile);
  const size_t lds = (size_t)a.pool_words * 4 + (a.k ? (size_t)a.chunks * 16 : 0);
  const bool small = a.n <= 4 && a.n_leaves <= 4 && a.n_req <= 4 && a.n_veto_all <= 4 && a.n_opt_all <= 4;
  const bool prox = a.prox_filter || (a.prox_slop && a.top_n);
#define RSGPU_HYBT(T, M)                                                                                                          \
  do {                                                                                                                            \
    if (a.P.n_nodes > 0 && prox) /* (round 6: the root's window / the per-hit slop over nested children: their leaves, flattened) */ \
      hipLaunchKernelGGL((hybrid_tree_tile_kernel<T, M, kHybTreeMaxLists, true, true>), dim3(n_tiles), dim3(256), lds, s, a);    \
    else if (a.P.n_nodes > 0)                                                                                                     \
      hipLaunchKernelGGL((hybrid_tree_tile_kernel<T, M, kHybTreeMaxLists, true, false>), dim3(n_tiles), dim3(256), lds, s, a);   \
    else if (small && prox) hipLaunchKernelGGL((hybrid_tree_tile_kernel<T, M, 4, false, true>), dim3(n_tiles), dim3(256), lds, s, a); \
    else if (prox)                                                                                                                \
      hipLaunchKernelGGL((hybrid_tree_tile_kernel<T, M, kHybTreeMaxLists, false, true>), dim3(n_tiles), dim3(256), lds, s, a);   \
    else hipLaunchKernelGGL((hybrid_tree_tile_kernel<T, M, kHybTreeMaxLists, false, false>), dim3(n_tiles), dim3(256), lds, s, a); \
  } while (0)
  if (!a.k) RSGPU_HYBT(KT_F32, KM_IP);
  else if (type == KT_F32 && metric == KM_L2) RSGPU_HYBT(KT_F32, KM_L2);
  else if (type == KT_F32) RSGPU_HYBT(KT_F32, KM_IP);
  else if (type == KT_F16 && metric == KM_L2) RSGPU_HYBT(KT_F16, KM_L2);
  else if (type == KT_F16) RSGPU_HYBT(KT_F16, KM_IP);
  else if (metric == KM_L2) RSGPU_HYBT(KT_BF16, KM_L2);
  else RSGPU_HYBT(KT_BF16, KM_IP);
#undef RSGPU_HYBT
}
bool knn_chain_supported(int type, int metric, uint32_t stride16) {
  if (type != KT_F32 && type != KT_F16 && type != KT_BF16) return false;
  if (metric != KM_L2 && metric != KM_IP) return false;
  return stride16 >= 1 && stride16 <= (uint32_t)kHybMaxChunks;
}
void launch_knn_chain_min(const void *rows, size_t stride, int type, int metric, const uint32_t *first_rows, uint32_t m,
                          const uint32_t *m_dev, const LabelRows &L, const void *query, float *dists, hipStream_t s) {
  if (!m || !L.next) return;
  KnnChainArgs a;
  a.rows = rows;
  a.stride16 = a.chunks = (uint32_t)(stride / 16);
  const Shape sh = pick_shape(a.stride16);
  a.G = sh.G;
  a.ITERS = sh.ITERS;
  a.query = query;
  a.first_rows = first_rows;
  a.m = m;
  a.m_dev = m_dev;
  a.L = L;
  a.dists = dists;
  const uint32_t gpb = 256u / (uint32_t)a.G, need = (m + gpb - 1) / gpb;
  const dim3 grid(need < 2048 ? need : 2048);
  const size_t lds = (size_t)a.chunks * 16;
#define RSGPU_CH(T, M) hipLaunchKernelGGL((knn_chain_min_kernel<T, M>), grid, dim3(256), lds, s, a)
  if (type == KT_F32 && metric == KM_L2) RSGPU_CH(KT_F32, KM_L2);
  else if (type == KT_F32) RSGPU_CH(KT_F32, KM_IP);
  else if (type == KT_F16 && metric == KM_L2) RSGPU_CH(KT_F16, KM_L2);
  else if (type == KT_F16) RSGPU_CH(KT_F16, KM_IP);
  else if (metric == KM_L2) RSGPU_CH(KT_BF16, KM_L2);
  else RSGPU_CH(KT_BF16, KM_IP);
#undef RSGPU_CH
}
void launch_hybrid_tie_ids(const uint64_t *skey, const uint32_t *sidx, uint32_t n, uint64_t tau, uint64_t *out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(hybrid_tie_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, s, skey, sidx, n, tau, out);
}
void launch_hybrid_hits_pack(const uint32_t *tile_hits, uint32_t n_tiles, int n_leaves, const uint32_t *src_ids,
                             const uint32_t *src_freqs, const uint32_t *src_epos, uint32_t src_stride, uint32_t *dst_ids,
                             uint32_t *dst_freqs, uint32_t *dst_epos, uint32_t dst_cap, uint32_t *total_out, hipStream_t s,
                             const HybridRuns *runs) {
  if (!n_tiles) return;
  HybridRuns r;
  memset(&r, 0, sizeof r);
  if (runs) r = *runs;
  hipLaunchKernelGGL(hybrid_hits_pack_kernel, dim3(n_tiles), dim3(256), 0, s, tile_hits, n_tiles, n_leaves, src_ids, src_freqs, src_epos,
                     src_stride, dst_ids, dst_freqs, dst_epos, dst_cap, total_out, r);
}
void launch_hybrid_hits_merge(const HybridRuns &runs, int n_leaves, const uint32_t *src_ids, const uint32_t *src_freqs,
                              const uint32_t *src_epos, uint32_t src_cap, uint32_t *dst_ids, uint32_t *dst_freqs, uint32_t *dst_epos,
                              uint32_t dst_cap, uint32_t max_total, hipStream_t s) {
  if (!runs.n || !max_total) return;
  const uint32_t need = (max_total + 255) / 256;
  hipLaunchKernelGGL(hybrid_hits_merge_kernel, dim3(need < 16384 ? need : 16384), dim3(256), 0, s, runs, n_leaves, src_ids, src_freqs,
                     src_epos, src_cap, dst_ids, dst_freqs, dst_epos, dst_cap);
}
HybridTileLite hybrid_tile_lite(const HybridTileArgs &a, size_t *lds_out) {
  HybridTileLite t;
  memset(&t, 0, sizeof t);
  t.n = a.n;
  for (int l = 0; l < kHybMaxLists; l++) {
    t.ids[l] = a.ids[l];
    t.freq[l] = a.freq[l];
    t.len[l] = a.len[l];
    t.add[l] = a.add[l];
    t.dir[l] = a.dir[l];
    t.dir_shift[l] = a.dir_shift[l];
    t.dir_n[l] = a.dir_n[l];
    t.P.idf[l] = a.P.idf[l];
    t.P.bm25_idf[l] = a.P.bm25_idf[l];
    t.P.weight[l] = a.P.weight[l];
  }
  t.top_n = a.top_n;
  t.P.scorer = a.P.scorer;
  t.P.n_groups = a.P.n_groups;
  t.P.avg_doc_len = a.P.avg_doc_len;
  t.P.root_weight = a.P.root_weight;
  t.P.min_score = a.P.min_score;
  t.P.inv_tanh = a.P.inv_tanh;
  t.P.slop = a.P.slop;
  t.P.table_off = a.P.table_off;
  t.doc_len = a.doc_len;
  t.doc_score = a.doc_score;
  t.max_freq = a.max_freq;
  t.table_n = a.table_n;
  t.k = a.k;
  t.rows = a.rows;
  t.stride16 = a.stride16;
  t.chunks = a.chunks;
  if (a.k) {
    const Shape sh = pick_shape(a.stride16);
    t.G = sh.G;
    t.ITERS = sh.ITERS;
  } else {
    t.G = 1;
    t.ITERS = 0;
  }
  t.query = a.query;
  t.ids_base = a.ids_base;
  t.L = a.L;
  t.tile_hits = a.tile_hits;
  t.part_skey = a.part_skey;
  t.part_sidx = a.part_sidx;
  t.part_knn = a.part_knn;
  t.pool_words = std::max<uint32_t>(4096u, (uint32_t)(a.n + 1) * kHybTile);
  t.len_score = a.len_score;
  t.knn_pipeline = a.knn_pipeline;
  if (lds_out) *lds_out = (size_t)t.pool_words * 4 + (a.k ? (size_t)a.chunks * 16 : 0);
  return t;
}
bool launch_hybrid_tiles_batch(HybridTileBatch &b, int type, int metric, size_t lds, hipStream_t s) {
  if (b.n_q < 1 || b.n_q > (uint32_t)kHybBatchMax) return false;
  bool any_k = false;
  for (uint32_t i = 0; i < b.n_q; i++) any_k |= b.q[i].k != 0;
  uint32_t blocks = 0, prev = 0;
  for (uint32_t j = 0; j < b.n_q; j++) {  // (ascending tile counts: the caller sorted)
    if (b.n_tiles[j] < prev || !b.n_tiles[j]) return false;
    blocks += b.interleave ? (b.n_tiles[j] - prev) * (b.n_q - j) : b.n_tiles[j];
    b.tile_end[j] = blocks;
    prev = b.n_tiles[j];
  }
#define RSGPU_HYBB(T, M) hipLaunchKernelGGL((hybrid_tile_batch_kernel<T, M, kHybDpt, kHybMaxLists>), dim3(blocks), dim3(256), lds, s, b)
  if (!any_k) RSGPU_HYBB(KT_F32, KM_IP);
  else if (type == KT_F32 && metric == KM_L2) RSGPU_HYBB(KT_F32, KM_L2);
  else if (type == KT_F32) RSGPU_HYBB(KT_F32, KM_IP);
  else if (type == KT_F16 && metric == KM_L2) RSGPU_HYBB(KT_F16, KM_L2);
  else if (type == KT_F16) RSGPU_HYBB(KT_F16, KM_IP);
  else if (metric == KM_L2) RSGPU_HYBB(KT_BF16, KM_L2);
  else RSGPU_HYBB(KT_BF16, KM_IP);
#undef RSGPU_HYBB
  return true;
}
void launch_hybrid_reduce_batch(const HybridReduceBatch &r, hipStream_t s) {
  hipLaunchKernelGGL(hybrid_reduce_batch_kernel, dim3(3 * r.n_q), dim3(1024), 0, s, r);
}
void launch_hybrid_reduce(const HybridReduceArgs &r, hipStream_t s) {
  hipLaunchKernelGGL(hybrid_reduce_kernel, dim3((r.k ? 1 : 0) + (r.top_n ? 1 : 0) + 1), dim3(1024), 0, s, r);
}

}  // namespace rsgpu
