// gemm_qs_kernels.hip -- "query-stationary" batched FLAT filter pass on the gfx950 matrix cores.
//
// Step 2 of the batched-query path (gemm_kernels.hip header): S[256 x N] = Q[256 x dim] * X^T, keep
// (row, distance) with distance 1 - S <= tau[q].  The tiled GEMM in gemm_kernels.hip moves BOTH operands
// through LDS, re-fills its ring at every 128-row tile and needs one 16-byte LDS read per MFMA.  Here the
// 256 queries never leave the register file:
//   * wave w keeps its queries for the WHOLE K = dim as MFMA B fragments (N side): 32 queries = 192
//     registers at dim 768.  Default: 8 waves x 32 queries, two waves per SIMD (256 registers each), so
//     that while one wave pays the ~100-200 cycles an LDS-DMA piece costs at issue
//     (MI355X_MICROARCH.md) or runs its filter epilogue, its partner keeps the SIMD's matrix pipe fed;
//     gemm_qs=2 selects 4 waves x 64 queries (one 512-register wave per SIMD, half the LDS reads);
//   * only the corpus streams: tiles of 32 rows x dim (48 KiB at 768 fp16) go global -> LDS by DMA
//     (global_load_lds_dwordx4, fully coalesced 1 KiB pieces, nontemporal, SGPR tile base + a per-lane
//     offset that never changes) into a ring that spans tile boundaries; the pieces of tile i+NS-1 are
//     issued one at a time inside the MFMA stream of tile i;
//   * one s_barrier per tile with a counted s_waitcnt vmcnt(N): DMAs of the younger tiles stay in flight;
//   * corpus fragments are read with inline-asm ds_read_b128 + counted lgkmcnt waits, PF fragments ahead,
//     pinned by sched_barrier (hipcc would sink each read next to its MFMA, wait lgkmcnt(0) every time and
//     drain vmcnt before any LDS load it can see while DMAs are pending);
//   * with the queries on N, all 16 accumulators of a lane belong to ONE query (C col = lane & 31): the
//     filter is 8 v_max3 + one compare per tile and lane against a per-lane threshold register, and a
//     lane's hits go to a sub-list only that lane writes -- register cursor, no atomics, nothing in the
//     epilogue waits on vmcnt; compact_cand_kernel concatenates the sub-lists per query afterwards.
// LDS layout of a tile: row-major, the 16-byte chunk index XOR-ed with (row & 15) so that the fragment
// read (32 rows, same chunk) is bank-conflict free (SQ_LDS_BANK_CONFLICT = 0); the same XOR picks the
// SOURCE chunk of each DMA lane (the DMA destination is lane-linear).
// L2 indexes (template L2, round 3): |x - q|^2 = 2 (|q|^2/2 + |x|^2/2 - x.q).  The rows' half norms hn[row] (fp32, an
// array next to the corpus) ride along with the tiles: wave 0 issues one more DMA piece per tile -- 64 floats, the first 32
// are the tile's -- into a side ring behind the corpus ring; after the MFMA stream a lane reads the norms of its 16 rows with
// four ds_read_b128 (issued right behind the last MFMA: their latency overlaps the matrix pipe's drain),
// subtracts them from its accumulators and runs the same max3 filter against hq[q] - tau[q]/2.  Emitted: 2 (hq - (s - hn)).
// The host hands in norms and hq SHRUNK by half the relative error band, which makes the emitted value a lower bound of the
// exact distance and the test a test of that lower bound; compact_cand_kernel turns it into the upper bound the thresholds
// are selected from (a per-row band: one huge row does not widen anybody else's), survivors are re-scored exactly.
// Measured on 10M x 768 fp16, batch 256 (profiles/r01_batch_qs.txt): 4.0 ms per pass vs 6.9 ms for the tiled
// kernel; the matrix pipe is busy 62 % of the cycles but the chip clocks down to ~1.4 GHz under the combined
// MFMA + LDS + 3.6 TB/s HBM load (power), which is what now bounds it.
#include <hip/hip_runtime.h>

#include "h8_quant.hpp"
#include "kernels.hpp"

namespace rsgpu {
namespace {

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// accumulator of the pass: fp32 for the half types, exact 32-bit integers for the int8 shadow (KT_I8)
template <int DT>
struct AccT {
  typedef f32x16 t;
};
template <>
struct AccT<KT_I8> {
  typedef i32x16 t;
};

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xFFFFFFFFu;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int DT>
__device__ __forceinline__ typename AccT<DT>::t mfma(u4 a, u4 b, typename AccT<DT>::t c) {
  if constexpr (DT == KT_F16) {
    half8 x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
  } else if constexpr (DT == KT_I8) {  // 32 int8 of K per instruction: the same 16 bytes per lane, twice the rate
    i32x4 x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, c, 0, 0, 0);
  } else {
    bf16x8 x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  }
}

struct QsArgs {
  const u4 *rows;     // corpus, row-contiguous, 2*KS chunks per row
  const u4 *queries;  // [256][2*KS chunks]
  uint32_t row_begin, row_end;
  const float *tau;     // [256]
  uint32_t *sub_count;  // [gridDim.x][256][2]
  uint2 *sub_cand;      // [gridDim.x][256][2][sub_cap] (row, distance bits): one list per (workgroup, query, lane half)
  uint32_t sub_cap;
  const float *qscale;  // KT_I8: [256] distance = 1 - qscale[q] * (integer dot); row scale x query scale
  const float *hnorm;   // L2: |x|^2 / 2 per corpus row (readable up to row_end + 95), else unused
  const float *hq;      // L2: [256] |q|^2 / 2
  uint32_t inv_h2;      // SRC_H8 (fp16 rows quantised in flight, h8_quant.hpp): the fp16 inverse scale, twice
};

// same for a wave-uniform value: pinned to an SGPR (otherwise kernel arguments and gridDim are re-loaded with
// s_load + s_waitcnt lgkmcnt(0) wherever they are used, which also drains the LDS pipeline)
__device__ __forceinline__ uint32_t opaque_s(uint32_t v) {
  asm volatile("" : "+s"(v));
  return v;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}

// value -> same value, opaque to the optimiser: stops loop-invariant address arithmetic from being hoisted
// out of the tile loop into (scarce) registers
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

// v_max3_f32 as is: fmaxf() would first canonicalise every operand (one more VALU op each); a NaN operand is
// ignored, which is what the filter wants (NaN distances never qualify)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float m;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  return m;
}

__device__ __forceinline__ int max3i(int a, int b, int c) {
  int m;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  return m;
}

// ds_read_b128 at addr + 256*o, o = 0..7 (a constant after unrolling: one case survives, the offset is an immediate)
__device__ __forceinline__ u4 lds_read16(uint32_t addr, int o) {
  u4 v;
  switch (o) {
    case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); break;
    case 1: asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(v) : "v"(addr)); break;
    case 2: asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(v) : "v"(addr)); break;
    case 3: asm volatile("ds_read_b128 %0, %1 offset:768" : "=v"(v) : "v"(addr)); break;
    case 4: asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(v) : "v"(addr)); break;
    case 5: asm volatile("ds_read_b128 %0, %1 offset:1280" : "=v"(v) : "v"(addr)); break;
    case 6: asm volatile("ds_read_b128 %0, %1 offset:1536" : "=v"(v) : "v"(addr)); break;
    default: asm volatile("ds_read_b128 %0, %1 offset:1792" : "=v"(v) : "v"(addr)); break;
  }
  return v;
}
// n is a constant after unrolling: exactly one case survives
__device__ __forceinline__ void wait_lgkm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)"); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)"); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)"); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)"); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)"); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)"); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)"); break;
    case 7: asm volatile("s_waitcnt lgkmcnt(7)"); break;
    case 8: asm volatile("s_waitcnt lgkmcnt(8)"); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)"); break;
    case 10: asm volatile("s_waitcnt lgkmcnt(10)"); break;
    default: asm volatile("s_waitcnt lgkmcnt(11)"); break;
  }
}

// QB = 32-query blocks per wave: 2 -> 4 waves (one per SIMD, 512 registers), 1 -> 8 waves (two per SIMD, 256
// registers each: while one wave pays the ~100-200 cycles an LDS-DMA piece costs at issue, or runs its filter
// epilogue, its partner keeps the SIMD's matrix pipe busy -- MI355X_MICROARCH.md "Two waves per SIMD")
template <int DT, int KS, int NS, int QB, bool L2>
__global__ __launch_bounds__(512 / QB, 1) void gemm_qs_kernel(QsArgs g) {
  static_assert(!L2 || DT != KT_I8, "the int8 pass has no L2 epilogue");
  constexpr int NW = 8 / QB;        // waves per workgroup
  constexpr int ROWC = 2 * KS;     // 16-byte chunks per row
  constexpr int TILE = 32 * ROWC;  // chunks per tile of 32 corpus rows
  constexpr int PPW = KS / NW;     // 1 KiB DMA pieces per wave per tile (a tile is KS pieces)
  constexpr int NA = QB == 2 ? 64 : 32;      // query fragments pinned to the accumulation registers (half the register file)
  constexpr int NORMC = L2 ? 16 : 0;  // L2: 64 half norms (16 chunks) per ring slot behind the corpus ring
  __shared__ u4 smem[NS * TILE + NS * NORMC];  // the ring(s), nothing else: one object (see gemm_kernels.hip)
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t r = lane & 31, h = lane >> 5;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;

  // The wave's 64 queries as MFMA B fragments (N side): lane holds query 64w + 32nb + r, halves
  // 16ks + 8h .. +8.  With the queries on N, every accumulator of a lane belongs to ONE query per nb
  // (C col = lane&31), so the filter needs one threshold register per nb and a lane's hits go to a
  // sub-list only this lane writes: the cursor is a register, no atomics anywhere.
  u4 Q[QB][KS];
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) Q[nb][ks] = g.queries[(size_t)(32 * QB * w + 32 * nb + r) * ROWC + 2 * ks + h];
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      if (nb * KS + ks < NA) asm volatile("" : "+a"(Q[nb][ks]));  // the rest stays in VGPRs
    }
  // distance d = 1 - s <= tau, tested first on s with a margin that covers the rounding of both subtractions
  float tau[QB], thr[QB];
  float qsc[QB];   // KT_I8 only
  int ithr[QB];    // KT_I8 only: the same test on the integer dot product
  uint32_t cur[QB];
#pragma unroll
  for (int nb = 0; nb < QB; nb++) {
    cur[nb] = 0;
    tau[nb] = g.tau[32 * QB * w + 32 * nb + r];
    float u = 1.0f - tau[nb];
    float mag = fabsf(tau[nb]) + fabsf(u);
    if constexpr (L2) {  // 2 (hq - v) <= tau  <=>  v >= hq - tau / 2, v = s - hn
      qsc[nb] = g.hq[32 * QB * w + 32 * nb + r];
      u = qsc[nb] - 0.5f * tau[nb];
      mag = fabsf(qsc[nb]) + fabsf(0.5f * tau[nb]) + fabsf(u);
    }
    thr[nb] = tau[nb] == -__builtin_inff() ? __builtin_inff() : (tau[nb] == __builtin_inff() ? -__builtin_inff() : u - mag * 2.4e-7f);
    if constexpr (!L2) qsc[nb] = 1.0f;
    ithr[nb] = 0x7fffffff;
    if constexpr (DT == KT_I8) {
      qsc[nb] = g.qscale[32 * QB * w + 32 * nb + r];
      // 1 - qsc * acc <= tau  <=>  acc >= (1 - tau) / qsc; taken one unit (and the float roundings) lower: the exact
      // test on the distance follows for the lanes that pass
      const float lim = thr[nb] / qsc[nb] - 1.0f - fabsf(thr[nb] / qsc[nb]) * 2.4e-7f;
      ithr[nb] = !(qsc[nb] > 0.0f) || thr[nb] == __builtin_inff() || lim >= 2147483520.0f ? 0x7fffffff
                 : (lim <= -2147483520.0f ? (int)0x80000000 : (int)floorf(lim));
    }
  }

  const uint32_t n = g.row_end - g.row_begin;
  const uint32_t n_tiles = (n + 31) / 32;
  const uint32_t mine = n_tiles > blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const uint32_t row_first = opaque_s(g.row_begin + blockIdx.x * 32);  // first row of this workgroup's tile 0
  const uint32_t row_step = opaque_s(gridDim.x * 32);                  // ... and the distance to its next tile
  const uint32_t row_end = opaque_s(g.row_end);

  // DMA of piece p (64 consecutive LDS chunks, 1 KiB) of this workgroup's i-th tile into ring slot `stage`.
  // The lane's source offset inside a tile never changes (poff, bytes); tile base and LDS destination are
  // wave-uniform, so a piece costs a few scalar instructions plus the DMA itself (SGPR base + VGPR offset).
  const uint32_t wu = __builtin_amdgcn_readfirstlane(w);
  uint32_t poff[PPW];
#pragma unroll
  for (int p = 0; p < PPW; p++) {
    const uint32_t s = 64 * (PPW * w + p) + lane;
    const uint32_t rr = s / ROWC, cc = s - rr * ROWC;
    poff[p] = 16u * (rr * ROWC + (cc ^ (rr & 15u)));
  }
  auto issue_piece = [&](uint32_t i, uint32_t stage, int p) {
    const uint32_t row0 = row_first + i * row_step;
    __attribute__((address_space(3))) void *lp =
        (__attribute__((address_space(3))) void *)(smem + stage * TILE + 64 * (PPW * wu + p));
    // a ragged last tile reads up to 31 rows past row_end: the caller guarantees they are allocated
    // (launch_gemm_qs contract); they are multiplied but never emitted
    const char *tile = reinterpret_cast<const char *>(g.rows) + (size_t)row0 * (ROWC * 16);
    // (opaque: keeps the offset a 32-bit register instead of a hoisted 64-bit pair)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tile + opaque(poff[p])), lp, 16, 0,
                                     2);  // nt: corpus lines are used once
  };
  // L2: the tile's half norms, one 256-byte piece (64 floats from the tile's first row on; the tile owns the first 32)
  // into the side ring.  Wave 0 only: its vmcnt ladder below counts PPW + 1 loads per tile.
  auto issue_norms = [&](uint32_t i, uint32_t stage) {
    if constexpr (L2) {
      if (wu == 0) {
        const uint32_t row0 = row_first + i * row_step;
        __attribute__((address_space(3))) void *lp = (__attribute__((address_space(3))) void *)(smem + NS * TILE + stage * NORMC);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g.hnorm + row0 + opaque(lane)), lp, 4, 0, 0);
      }
    }
  };
  auto issue = [&](uint32_t i, uint32_t stage) {
    issue_norms(i, stage);
#pragma unroll
    for (int p = 0; p < PPW; p++) issue_piece(i, stage, p);
  };

  uint32_t fill = 0;  // ring slot of the next tile to issue
#pragma unroll
  for (int p = 0; p < NS - 1; p++)
    if ((uint32_t)p < mine) {
      issue(p, fill);
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
  uint32_t stage = 0;
  for (uint32_t i = 0; i < mine; i++) {
    // tile i has landed once only this wave's DMAs of the younger tiles are outstanding (loads return
    // in order; candidate stores in the queue can only make the wait conservative) ...
    const uint32_t younger = mine - 1 - i < (uint32_t)(NS - 2) ? mine - 1 - i : (uint32_t)(NS - 2);
    if (L2 && wu == 0) {  // (wave 0 of an L2 pass also has the tiles' norm pieces in its queue)
      switch (younger) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<PPW + 1>(); break;
        case 2: wait_vm<2 * (PPW + 1)>(); break;
        case 3: wait_vm<3 * (PPW + 1)>(); break;
        case 4: wait_vm<4 * (PPW + 1)>(); break;
        case 5: wait_vm<5 * (PPW + 1)>(); break;
        default: wait_vm<6 * (PPW + 1)>(); break;
      }
    } else {
      switch (younger) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<PPW>(); break;
        case 2: wait_vm<2 * PPW>(); break;
        case 3: wait_vm<3 * PPW>(); break;
        case 4: wait_vm<4 * PPW>(); break;
        case 5: wait_vm<5 * PPW>(); break;
        default: wait_vm<6 * PPW>(); break;
      }
    }
    // ... for every wave; the barrier also says tile i-1 has been consumed by all, freeing its slot
    __builtin_amdgcn_s_barrier();
    // the DMAs of tile i+NS-1 are issued one piece every fourth k-step INSIDE the MFMA stream below: a
    // burst here would block this wave -- the only one on its SIMD -- on the memory pipeline's back
    // pressure while the matrix core idles (measured: DMA time and MFMA time added up)
    const bool refill = i + NS - 1 < mine;
    // corpus fragment (ks, lane): row r, chunk (2ks+h) ^ (r&15); the XOR only touches the low four bits, so
    // eight lane addresses (ks & 7) plus an immediate 256-byte step per eight k-steps cover the whole row
    const uint32_t lr = opaque(lane);
    const uint32_t tbase = lds_base + 16u * (stage * TILE + (lr & 31u) * ROWC);
    const uint32_t nstage = opaque_s(stage);
    stage = stage + 1 == NS ? 0 : stage + 1;
    // chunk (2ks+h)^x = 16(ks>>3) + (2(ks&7) ^ (h^x)): one v_xad_u32 per read, nothing held in registers
    const uint32_t t16 = 16u * ((lr >> 5) ^ (lr & 15u));
    auto frag_addr = [&](int ks) { return tbase + ((32u * (uint32_t)(ks & 7)) ^ t16); };
    typename AccT<DT>::t acc[QB];
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
      for (int nb = 0; nb < QB; nb++) acc[nb][e] = 0;
    // Software pipeline, PF fragments ahead: with one wave per SIMD nothing else hides the LDS latency.
    // The reads and their counted waits are inline asm (hipcc waits lgkmcnt(0) before every use once LDS
    // DMAs are pending, and would drain vmcnt for an LDS load it can see); sched_barrier pins the order.
    constexpr int PF = QB == 1 ? 3 : 6;
    u4 xb[PF];
#pragma unroll
    for (int ks = 0; ks < PF; ks++) xb[ks] = lds_read16(frag_addr(ks), ks >> 3);
    u4 hnv[L2 ? 4 : 1];  // L2: half norms of the lane's 16 rows
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      // fragment ks has arrived
      wait_lgkm(KS - 1 - ks < PF - 1 ? KS - 1 - ks : PF - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < QB; nb++) acc[nb] = mfma<DT>(xb[ks % PF], Q[nb][ks], acc[nb]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + PF < KS) xb[ks % PF] = lds_read16(frag_addr(ks + PF), (ks + PF) >> 3);
      if constexpr (L2) {  // behind the last MFMA: the fragment registers are free, the reads overlap the pipe's drain
        if (ks == KS - 1) {
          // the norms of this lane's rows 8 eg + 4 h .. + 3 are chunk 2 eg + h of the slot's norm piece
          const uint32_t nbase = lds_base + 16u * (NS * TILE + nstage * NORMC + (opaque(lane) >> 5));
#pragma unroll
          for (int eg = 0; eg < 4; eg++) hnv[eg] = lds_read16(nbase + 32u * eg, 0);
        }
      }
      if (ks % (KS / PPW) == 1 && refill) {
        if (ks / (KS / PPW) == 0) issue_norms(i + NS - 1, fill);
        issue_piece(i + NS - 1, fill, ks / (KS / PPW));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (refill) fill = fill + 1 == NS ? 0 : fill + 1;
    if constexpr (L2) {  // v = s - hn: the filter below then runs on v against hq - tau / 2
      wait_lgkm(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < QB; nb++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[nb][e] -= __uint_as_float(hnv[e >> 2][e & 3]);
    }
    // C row = corpus row (e&3)+8*(e>>2)+4h of the tile, C col = query r of the nb block
    const uint32_t xr0 = row_first + i * row_step + 4 * (lr >> 5);
#pragma unroll
    for (int nb = 0; nb < QB; nb++) {
      if constexpr (DT == KT_I8) {
        int m = max3i(acc[nb][0], acc[nb][1], acc[nb][2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) m = max3i(m, acc[nb][e], acc[nb][e + 1]);
        m = max3i(m, acc[nb][15], acc[nb][15]);
        if (m >= ithr[nb]) {  // a few lanes per tile (the int8 band is wide: ~20 % of a wavefront's tiles at K = 100)
          uint2 *list = g.sub_cand + ((((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + (lr & 31u)) * 2 + (lr >> 5)) * g.sub_cap);
#pragma unroll
          for (int eg = 0; eg < 4; eg++) {
            const int gm = max3i(max3i(acc[nb][4 * eg], acc[nb][4 * eg + 1], acc[nb][4 * eg + 2]), acc[nb][4 * eg + 3], acc[nb][4 * eg + 3]);
            if (gm >= ithr[nb]) {
#pragma unroll
              for (int el = 0; el < 4; el++) {
                const float d = 1.0f - qsc[nb] * (float)acc[nb][4 * eg + el];
                const uint32_t xr = xr0 + 8 * eg + el;
                if (acc[nb][4 * eg + el] >= ithr[nb] && d <= tau[nb] && xr < row_end) {
                  if (cur[nb] < g.sub_cap) list[cur[nb]] = make_uint2(xr, __float_as_uint(d));
                  cur[nb]++;
                }
              }
            }
          }
        }
      } else {
        float m = max3(acc[nb][0], acc[nb][1], acc[nb][2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) m = max3(m, acc[nb][e], acc[nb][e + 1]);
        m = max3(m, acc[nb][15], acc[nb][15]);
        if (m >= thr[nb]) {  // rare: a few hits per tile over the whole wave
          uint2 *list = g.sub_cand + ((((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + (lr & 31u)) * 2 + (lr >> 5)) * g.sub_cap);
#pragma unroll
          for (int eg = 0; eg < 4; eg++) {
            const float gm = max3(max3(acc[nb][4 * eg], acc[nb][4 * eg + 1], acc[nb][4 * eg + 2]), acc[nb][4 * eg + 3], acc[nb][4 * eg + 3]);
            if (gm >= thr[nb]) {
#pragma unroll
              for (int el = 0; el < 4; el++) {
                const float d = L2 ? 2.0f * (qsc[nb] - acc[nb][4 * eg + el]) : 1.0f - acc[nb][4 * eg + el];
                const uint32_t xr = xr0 + 8 * eg + el;
                // (L2: the test on v IS the filter -- a superset of d <= tau by the margin in thr, which is all a filter
                // pass owes; tau itself is not kept in a register)
                if ((L2 ? acc[nb][4 * eg + el] >= thr[nb] : d <= tau[nb]) && xr < row_end) {
                  if (cur[nb] < g.sub_cap) list[cur[nb]] = make_uint2(xr, __float_as_uint(d));
                  cur[nb]++;
                }
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
    g.sub_count[((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + r) * 2 + h] = cur[nb];
}

// ---- FLOAT32 rows on the matrix cores, no stored shadow (round 4) --------------------------------------------------------
// The same query-stationary pass over the fp32 corpus itself: the tiles go global -> LDS by DMA as they are (fp32), and a
// wave turns the 32 bytes of a fragment (8 fp32 of one row) into the 16 bytes the MFMA wants with four v_cvt_pk_bf16_f32 on
// its way from LDS to the matrix pipe -- bf16 keeps fp32's exponent range, so no row can overflow or flush, whatever the
// metric.  HBM traffic is the 4 bytes per element the exact scan reads anyway; nothing is stored next to the index.  The
// pass is a FILTER: |x~.q~ - x.q| <= (2u + u^2) |x||q| with u = 2^-9 (both operands rounded to nearest) plus the two fp32
// summation orders; the host widens every threshold by that band and the survivors are re-scored from the same fp32 rows with
// the single-query scan's arithmetic (batch_rescore_kernel) -> ids and distances bit-identical to VecSimIndex_TopKQuery.
// A 32-row tile of fp32 rows is 96 KiB at dim 768, so the ring's slots hold a K-PART of a tile: 32 rows x KH k-steps (16
// elements each) = 2 KiB x KH (48 KiB at KH = 24); the accumulators live across the NPART = KS / KH parts of a tile and the
// filter epilogue runs behind the last one.  Fragment (ksl, lane): row r = lane & 31, chunks 4 ksl + 2 h + {0, 1} of the
// slot row, XOR-ed with (r & 15) as in the 16-bit kernel: a b128 read's sixteen-lane groups hold sixteen different r & 15,
// hence sixteen different chunks of one 256-byte bank line -- conflict free.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (one v_cvt_pk_bf16_f32, round to nearest even; left to the compiler so that it sees the instruction next to the MFMAs)
__device__ __forceinline__ uint32_t cvt_pk_bf16(uint32_t a, uint32_t b) {
  const f32x2 f = {__uint_as_float(a), __uint_as_float(b)};
  const bf16x2 v = __builtin_convertvector(f, bf16x2);
  uint32_t r;
  __builtin_memcpy(&r, &v, 4);
  return r;
}

// SRC_H8 (round 6): the SAME kernel over FLOAT16 rows, quantised to int8 on their way from LDS to the int8 matrix pipe
// (h8_quant.hpp: one v_pk_fma_f16 per two elements, one v_perm_b32 per four) -- a k-step is 32 elements = the same 64 bytes
// of a row, the ring / DMA / fragment geometry is the fp32 form's at half the dim, the instruction is v_mfma_i32_32x32x32_i8
// (half the matrix-pipe cycles of the fp16 form per row, which at configs[2] is what the chip's power budget pays for) and
// the filter is the int8 pass's integer compare.  Nothing is stored next to the index but four index-wide numbers.
enum : int { SRC_F32 = 0, SRC_H8 = 1 };
template <int KS, int KH, int NS, int QB, bool L2, int SRC = SRC_F32>
__global__ __launch_bounds__(512 / QB, 1) void gemm_qs_f32_kernel(QsArgs g) {
  static_assert(KS % KH == 0, "a tile is a whole number of K parts");
  static_assert(SRC == SRC_F32 || !L2, "the int8 form has no L2 epilogue");
  constexpr int ADT = SRC == SRC_H8 ? KT_I8 : KT_BF16;  // the matrix instruction's operand type
  constexpr int NW = 8 / QB;          // waves per workgroup
  constexpr int NPART = KS / KH;      // ring slots per 32-row tile
  constexpr int RC = 4 * KH;          // 16-byte chunks per slot row (fp32: four per k-step)
  constexpr int SROW = 4 * KS;        // ... per corpus row
  constexpr int SLOT = 32 * RC;       // chunks per slot
  constexpr int PPW = 2 * KH / NW;    // 1 KiB DMA pieces per wave per slot (a slot is 2 KH pieces)
  constexpr int STEP = KH / PPW;      // a piece is issued every STEP k-steps
  static_assert(PPW * NW == 2 * KH && STEP * PPW == KH && RC % 16 == 0, "shape");
  constexpr int NA = QB == 2 ? 64 : 32;
  constexpr int NORMC = L2 ? 16 : 0;
  __shared__ u4 smem[NS * SLOT + NS * NORMC];
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t r = lane & 31, h = lane >> 5;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;

  // the wave's queries (bf16, made by convert_queries_bf16_kernel) as B fragments for the whole K, as in gemm_qs_kernel
  u4 Q[QB][KS];
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) Q[nb][ks] = g.queries[(size_t)(32 * QB * w + 32 * nb + r) * (2 * KS) + 2 * ks + h];
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      if (nb * KS + ks < NA) asm volatile("" : "+a"(Q[nb][ks]));
    }
  float tau[QB], thr[QB], qsc[QB];
  int ithr[QB];  // SRC_H8 only
  uint32_t cur[QB];
#pragma unroll
  for (int nb = 0; nb < QB; nb++) {
    cur[nb] = 0;
    tau[nb] = g.tau[32 * QB * w + 32 * nb + r];
    float u = 1.0f - tau[nb];
    float mag = fabsf(tau[nb]) + fabsf(u);
    qsc[nb] = 1.0f;
    if constexpr (L2) {
      qsc[nb] = g.hq[32 * QB * w + 32 * nb + r];
      u = qsc[nb] - 0.5f * tau[nb];
      mag = fabsf(qsc[nb]) + fabsf(0.5f * tau[nb]) + fabsf(u);
    }
    thr[nb] = tau[nb] == -__builtin_inff() ? __builtin_inff() : (tau[nb] == __builtin_inff() ? -__builtin_inff() : u - mag * 2.4e-7f);
    ithr[nb] = 0x7fffffff;
    if constexpr (SRC == SRC_H8) {  // the int8 pass's test on the integer dot product (gemm_qs_kernel, KT_I8)
      qsc[nb] = g.qscale[32 * QB * w + 32 * nb + r];
      const float lim = thr[nb] / qsc[nb] - 1.0f - fabsf(thr[nb] / qsc[nb]) * 2.4e-7f;
      ithr[nb] = !(qsc[nb] > 0.0f) || thr[nb] == __builtin_inff() || lim >= 2147483520.0f ? 0x7fffffff
                 : (lim <= -2147483520.0f ? (int)0x80000000 : (int)floorf(lim));
    }
  }
  const uint32_t inv2 = opaque_s(g.inv_h2);

  const uint32_t n = g.row_end - g.row_begin;
  const uint32_t n_tiles = (n + 31) / 32;
  const uint32_t mine = n_tiles > blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const uint32_t n_slots = mine * NPART;
  const uint32_t row_first = opaque_s(g.row_begin + blockIdx.x * 32);
  const uint32_t row_step = opaque_s(gridDim.x * 32);
  const uint32_t row_end = opaque_s(g.row_end);

  const uint32_t wu = __builtin_amdgcn_readfirstlane(w);
  uint32_t poff[PPW];  // the lane's source offset inside a (tile, part): bytes from the part's first byte of the tile's first row
#pragma unroll
  for (int p = 0; p < PPW; p++) {
    const uint32_t s = 64 * (PPW * w + p) + lane;
    const uint32_t rr = s / RC, cc = s - rr * RC;
    poff[p] = 16u * (rr * SROW + (cc ^ (rr & 15u)));
  }
  // slot j = (tile j / NPART, part j % NPART) of this workgroup
  auto issue_piece = [&](uint32_t j, uint32_t stage, int p) {
    const uint32_t i = NPART == 1 ? j : j / NPART, part = NPART == 1 ? 0 : j - i * NPART;
    const uint32_t row0 = row_first + i * row_step;
    __attribute__((address_space(3))) void *lp =
        (__attribute__((address_space(3))) void *)(smem + stage * SLOT + 64 * (PPW * wu + p));
    const char *src = reinterpret_cast<const char *>(g.rows) + (size_t)row0 * (SROW * 16) + part * (RC * 16);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + opaque(poff[p])), lp, 16, 0, 2);
  };
  // L2: the tile's half norms ride with EVERY slot of the tile (256 bytes next to 2 KiB x KH: the ladder below stays one
  // count per slot); the epilogue reads them from the last part's stage
  auto issue_norms = [&](uint32_t j, uint32_t stage) {
    if constexpr (L2) {
      if (wu == 0) {
        const uint32_t i = NPART == 1 ? j : j / NPART;
        const uint32_t row0 = row_first + i * row_step;
        __attribute__((address_space(3))) void *lp = (__attribute__((address_space(3))) void *)(smem + NS * SLOT + stage * NORMC);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g.hnorm + row0 + opaque(lane)), lp, 4, 0, 0);
      }
    }
  };
  auto issue = [&](uint32_t j, uint32_t stage) {
    issue_norms(j, stage);
#pragma unroll
    for (int p = 0; p < PPW; p++) issue_piece(j, stage, p);
  };

  uint32_t fill = 0;
#pragma unroll
  for (int p = 0; p < NS - 1; p++)
    if ((uint32_t)p < n_slots) {
      issue(p, fill);
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
  uint32_t stage = 0;
  for (uint32_t i = 0; i < mine; i++) {
    typename AccT<ADT>::t acc[QB];
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
      for (int nb = 0; nb < QB; nb++) acc[nb][e] = 0;
    u4 hnv[L2 ? 4 : 1];
#pragma unroll
    for (int part = 0; part < NPART; part++) {
      const uint32_t j = i * NPART + part;
      const uint32_t younger = n_slots - 1 - j < (uint32_t)(NS - 2) ? n_slots - 1 - j : (uint32_t)(NS - 2);
      if (L2 && wu == 0) {
        switch (younger) {
          case 0: wait_vm<0>(); break;
          case 1: wait_vm<PPW + 1>(); break;
          case 2: wait_vm<2 * (PPW + 1)>(); break;
          case 3: wait_vm<3 * (PPW + 1)>(); break;
          case 4: wait_vm<4 * (PPW + 1)>(); break;
          case 5: wait_vm<5 * (PPW + 1)>(); break;
          default: wait_vm<6 * (PPW + 1)>(); break;
        }
      } else {
        switch (younger) {
          case 0: wait_vm<0>(); break;
          case 1: wait_vm<PPW>(); break;
          case 2: wait_vm<2 * PPW>(); break;
          case 3: wait_vm<3 * PPW>(); break;
          case 4: wait_vm<4 * PPW>(); break;
          case 5: wait_vm<5 * PPW>(); break;
          default: wait_vm<6 * PPW>(); break;
        }
      }
      __builtin_amdgcn_s_barrier();
      const bool refill = j + NS - 1 < n_slots;
      // (opaque: the eight fragment addresses of a slot are one v_xad_u32 each, not eight registers held across the loop)
      const uint32_t lr = opaque(lane);
      // chunk (4 ksl + 2 h + e) ^ (r & 15) = 16 (ksl >> 2) + ((4 (ksl & 3) + e) ^ (2 h ^ (r & 15)))
      const uint32_t t16 = 16u * ((2u * (lr >> 5)) ^ (lr & 15u));
      const uint32_t tbase = lds_base + 16u * (stage * SLOT + (lr & 31u) * RC);
      const uint32_t nstage = opaque_s(stage);
      stage = stage + 1 == NS ? 0 : stage + 1;
      auto frag_addr = [&](int ksl, int e) { return tbase + ((64u * (uint32_t)(ksl & 3) + 16u * (uint32_t)e) ^ t16); };
      constexpr int PF = QB == 1 ? 2 : 3;  // fragments (two reads each) ahead
      u4 xa[PF], xb[PF];
#pragma unroll
      for (int ksl = 0; ksl < PF && ksl < KH; ksl++) {
        xa[ksl] = lds_read16(frag_addr(ksl, 0), ksl >> 2);
        xb[ksl] = lds_read16(frag_addr(ksl, 1), ksl >> 2);
      }
#pragma unroll
      for (int ksl = 0; ksl < KH; ksl++) {
        const int ahead = KH - 1 - ksl < PF - 1 ? KH - 1 - ksl : PF - 1;
        wait_lgkm(2 * ahead);
        __builtin_amdgcn_sched_barrier(0);
        u4 f;
        if constexpr (SRC == SRC_H8) {  // 16 fp16 of the row -> 16 int8 (element order kept: the queries' int8 rows are plain)
          f[0] = h8_pack4(h8_quant2(xa[ksl % PF][0], inv2), h8_quant2(xa[ksl % PF][1], inv2));
          f[1] = h8_pack4(h8_quant2(xa[ksl % PF][2], inv2), h8_quant2(xa[ksl % PF][3], inv2));
          f[2] = h8_pack4(h8_quant2(xb[ksl % PF][0], inv2), h8_quant2(xb[ksl % PF][1], inv2));
          f[3] = h8_pack4(h8_quant2(xb[ksl % PF][2], inv2), h8_quant2(xb[ksl % PF][3], inv2));
        } else {
          f[0] = cvt_pk_bf16(xa[ksl % PF][0], xa[ksl % PF][1]);
          f[1] = cvt_pk_bf16(xa[ksl % PF][2], xa[ksl % PF][3]);
          f[2] = cvt_pk_bf16(xb[ksl % PF][0], xb[ksl % PF][1]);
          f[3] = cvt_pk_bf16(xb[ksl % PF][2], xb[ksl % PF][3]);
        }
#pragma unroll
        for (int nb = 0; nb < QB; nb++) acc[nb] = mfma<ADT>(f, Q[nb][part * KH + ksl], acc[nb]);
        __builtin_amdgcn_sched_barrier(0);
        if (ksl + PF < KH) {
          xa[ksl % PF] = lds_read16(frag_addr(ksl + PF, 0), (ksl + PF) >> 2);
          xb[ksl % PF] = lds_read16(frag_addr(ksl + PF, 1), (ksl + PF) >> 2);
        }
        if constexpr (L2) {
          if (part == NPART - 1 && ksl == KH - 1) {
            const uint32_t nbase = lds_base + 16u * (NS * SLOT + nstage * NORMC + (opaque(lane) >> 5));
#pragma unroll
            for (int eg = 0; eg < 4; eg++) hnv[eg] = lds_read16(nbase + 32u * eg, 0);
          }
        }
        if (ksl % STEP == (STEP > 1 ? 1 : 0) && refill) {
          if (ksl / STEP == 0) issue_norms(j + NS - 1, fill);
          issue_piece(j + NS - 1, fill, ksl / STEP);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (refill) fill = fill + 1 == NS ? 0 : fill + 1;
    }
    if constexpr (L2) {
      wait_lgkm(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < QB; nb++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[nb][e] -= __uint_as_float(hnv[e >> 2][e & 3]);
    }
    const uint32_t lr = opaque(lane);
    const uint32_t xr0 = row_first + i * row_step + 4 * (lr >> 5);
#pragma unroll
    for (int nb = 0; nb < QB; nb++) {
      if constexpr (SRC == SRC_H8) {  // the int8 pass's epilogue (gemm_qs_kernel, KT_I8)
        int m = max3i(acc[nb][0], acc[nb][1], acc[nb][2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) m = max3i(m, acc[nb][e], acc[nb][e + 1]);
        m = max3i(m, acc[nb][15], acc[nb][15]);
        if (m >= ithr[nb]) {
          uint2 *list = g.sub_cand + ((((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + (lr & 31u)) * 2 + (lr >> 5)) * g.sub_cap);
#pragma unroll
          for (int eg = 0; eg < 4; eg++) {
            const int gm = max3i(max3i(acc[nb][4 * eg], acc[nb][4 * eg + 1], acc[nb][4 * eg + 2]), acc[nb][4 * eg + 3], acc[nb][4 * eg + 3]);
            if (gm >= ithr[nb]) {
#pragma unroll
              for (int el = 0; el < 4; el++) {
                const float d = 1.0f - qsc[nb] * (float)acc[nb][4 * eg + el];
                const uint32_t xr = xr0 + 8 * eg + el;
                if (acc[nb][4 * eg + el] >= ithr[nb] && d <= tau[nb] && xr < row_end) {
                  if (cur[nb] < g.sub_cap) list[cur[nb]] = make_uint2(xr, __float_as_uint(d));
                  cur[nb]++;
                }
              }
            }
          }
        }
      } else {
        float m = max3(acc[nb][0], acc[nb][1], acc[nb][2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) m = max3(m, acc[nb][e], acc[nb][e + 1]);
        m = max3(m, acc[nb][15], acc[nb][15]);
        if (m >= thr[nb]) {
          uint2 *list = g.sub_cand + ((((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + (lr & 31u)) * 2 + (lr >> 5)) * g.sub_cap);
#pragma unroll
          for (int eg = 0; eg < 4; eg++) {
            const float gm = max3(max3(acc[nb][4 * eg], acc[nb][4 * eg + 1], acc[nb][4 * eg + 2]), acc[nb][4 * eg + 3], acc[nb][4 * eg + 3]);
            if (gm >= thr[nb]) {
#pragma unroll
              for (int el = 0; el < 4; el++) {
                const float d = L2 ? 2.0f * (qsc[nb] - acc[nb][4 * eg + el]) : 1.0f - acc[nb][4 * eg + el];
                const uint32_t xr = xr0 + 8 * eg + el;
                // (the test on the accumulator IS the filter -- a superset of d <= tau by the margin in thr, which is all a
                // filter pass owes: the survivors are re-scored; tau itself is not kept in a register)
                if (acc[nb][4 * eg + el] >= thr[nb] && xr < row_end) {
                  if (cur[nb] < g.sub_cap) list[cur[nb]] = make_uint2(xr, __float_as_uint(d));
                  cur[nb]++;
                }
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
    g.sub_count[((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + r) * 2 + h] = cur[nb];
}

// ---- FLOAT16 rows -> int8 ONCE per workgroup, register-staged (round 6, gemm_qs_h8=5/6) ---------------------------------------
// gemm_qs_f32_kernel<.., SRC_H8> quantises every fragment in every wave: 12 VALU per k-step per wave, 436 VALU per tile and wave --
// its waves spend half their cycles issuing and a third parked at the per-slot barrier (profiles/r06_batch_qs_pmc_h8.json).  Here a
// 32-row tile is quantised ONCE: the fp16 chunks travel global -> REGISTERS (plain nontemporal 16-byte loads, D tiles in flight per
// thread: the registers are the ring), each thread turns its chunks into 8 int8 bytes (4 v_pk_fma_f16 + 2 v_perm_b32, h8_quant.hpp)
// and stores them into a double-buffered int8 tile in LDS (24 KiB at dim 768, rows XOR-swizzled as everywhere in this file); every
// wave then reads plain int8 fragments -- one ds_read_b128 per matrix instruction, no VALU in the matrix stream -- against its
// register-stationary int8 queries.  One barrier per tile; eight waves (two per SIMD) so that one wave's barrier / memory waits are the
// other's matrix time.  LDS traffic per tile: 24 KiB written + NW x 24 KiB read (the DMA form: 48 written + NW x 48 read).
// SRC_F8 (fp32 rows, the RediSearch default type): the same kernel -- a 16-byte chunk holds four elements instead of eight and becomes
// one dword of the int8 tile (f8_quant4: four v_fma_f32 + three v_perm_b32); twice the chunks per tile and thread.
enum : int { SRC_F8 = 2 };
template <int KS, int QB, int D, int SRC = SRC_H8>
__global__ __launch_bounds__(512 / QB, 1) void gemm_qs_h8r_kernel(QsArgs g) {
  constexpr int NW = 8 / QB, NT = 64 * NW;
  constexpr int FC = (SRC == SRC_F8 ? 8 : 4) * KS;  // source chunks per row (a k-step is 32 elements)
  constexpr int CPT = 32 * FC / NT;      // chunks per thread and tile
  static_assert(CPT * NT == 32 * FC, "a tile is a whole number of chunks per thread");
  constexpr int RS = (2 * KS + 15) / 16 * 16;  // int8 row stride in 16-byte chunks (the swizzle needs whole groups of 16)
  constexpr int TILE = 32 * RS;                // chunks per int8 tile
  __shared__ u4 smem[2 * TILE];
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t r = lane & 31, h = lane >> 5;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;

  u4 Q[QB][KS];
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) Q[nb][ks] = g.queries[(size_t)(32 * QB * w + 32 * nb + r) * (2 * KS) + 2 * ks + h];
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) asm volatile("" : "+a"(Q[nb][ks]));  // B operands straight from the accumulation registers
  float tau[QB], qsc[QB];
  int ithr[QB];
  uint32_t cur[QB];
#pragma unroll
  for (int nb = 0; nb < QB; nb++) {
    cur[nb] = 0;
    tau[nb] = g.tau[32 * QB * w + 32 * nb + r];
    const float u = 1.0f - tau[nb], mag = fabsf(tau[nb]) + fabsf(u);
    const float thr = tau[nb] == -__builtin_inff() ? __builtin_inff() : (tau[nb] == __builtin_inff() ? -__builtin_inff() : u - mag * 2.4e-7f);
    qsc[nb] = g.qscale[32 * QB * w + 32 * nb + r];
    const float lim = thr / qsc[nb] - 1.0f - fabsf(thr / qsc[nb]) * 2.4e-7f;
    ithr[nb] = !(qsc[nb] > 0.0f) || thr == __builtin_inff() || lim >= 2147483520.0f ? 0x7fffffff
               : (lim <= -2147483520.0f ? (int)0x80000000 : (int)floorf(lim));
  }
  const uint32_t inv2 = g.inv_h2;
  const float invf = __uint_as_float(g.inv_h2);  // SRC_F8: the fp32 inverse scale travels in the same field

  const uint32_t n = g.row_end - g.row_begin;
  const uint32_t n_tiles = (n + 31) / 32;
  const uint32_t mine = n_tiles > blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const uint32_t row_first = g.row_begin + blockIdx.x * 32, row_step = gridDim.x * 32, row_end = g.row_end;

  // chunk c = tid + NT j of a tile: row c / FC, source chunk c % FC -> fp16: the 8 bytes at int8 chunk (cc / 2) ^ (row & 15), half
  // cc & 1; fp32: the 4 bytes at int8 chunk (cc / 4) ^ (row & 15), quarter cc & 3.  The fp16 form keeps the CPT offsets in
  // registers; the fp32 form (twice the chunks, twice the ring) recomputes them -- FC = 2^s or 3 * 2^s: a shift and, for the
  // factor three, (x * 171) >> 9 (exact below 384) -- the twelve registers are the difference between fitting 256 and spilling.
  constexpr bool kKeepOff = SRC != SRC_F8;
  constexpr int FS = (FC % 3 == 0) ? __builtin_ctz(FC / 3) : __builtin_ctz(FC);
  auto off_of = [&](int j) {
    const uint32_t c = tid + NT * j;
    const uint32_t rr = (FC % 3 == 0) ? (((c >> FS) * 171u) >> 9) : (c >> FS), cc = c - rr * FC;
    return SRC == SRC_F8 ? 16u * (rr * RS + ((cc >> 2) ^ (rr & 15u))) + 4u * (cc & 3u)
                         : 16u * (rr * RS + ((cc >> 1) ^ (rr & 15u))) + 8u * (cc & 1u);
  };
  uint32_t woff[kKeepOff ? CPT : 1];
  if constexpr (kKeepOff) {
#pragma unroll
    for (int j = 0; j < CPT; j++) woff[j] = off_of(j);
  }
  // (a ragged last tile reads up to 31 rows past row_end: allocated by the launch_gemm_qs contract, multiplied, never emitted)
  // a tile's first byte (wave-uniform: an SGPR pair); the thread's chunk j of it lies 16 (tid + NT j) bytes behind -- SGPR base +
  // 32-bit VGPR offset, so that the CPT addresses of a tile cost one register and an add each, not a 64-bit pair each
  auto tile_src = [&](uint32_t i) { return g.rows + (size_t)(row_first + i * row_step) * FC; };
  const uint32_t toff = 16u * tid;
  auto quant_store = [&](const u4 &x, uint32_t buf, int j) {
    if constexpr (SRC == SRC_F8) {
      *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(smem) + buf * (TILE * 16) + off_of(j)) = f8_quant4(x, invf);
    } else {
      const uint32_t t0 = h8_quant2(x[0], inv2), t1 = h8_quant2(x[1], inv2), t2 = h8_quant2(x[2], inv2), t3 = h8_quant2(x[3], inv2);
      *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(smem) + buf * (TILE * 16) + woff[j]) = make_uint2(h8_pack4(t0, t1), h8_pack4(t2, t3));
    }
  };

  // The ring: tiles i+1 .. i+D of this workgroup, on their way from HBM.  Loads and their waits are inline asm: hipcc's own
  // vmcnt bookkeeping allowed CPT - 1 loads in flight where D CPT - 1 are (it drained a whole register set per chunk: 3.35 ms
  // per pass against 2.9 with the counts below).  ld() requests, use() returns the register once only the D CPT - 1 requests
  // issued after it may still be outstanding -- requests retire in order; the candidate stores of the epilogue (asm as well)
  // can only make the wait conservative.  The "+v" operand ties every consumer to its wait.
  u4 R[D][CPT];
  auto ld = [&](u4 &dst, const u4 *tile, int j) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(dst) : "v"(toff + 16u * NT * (uint32_t)j), "s"(tile) : "memory");
  };
  auto use = [&](u4 &reg) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(D * CPT - 1) : "memory"); };
  // prologue: tile 0 straight through, tiles 1 .. D requested (a workgroup with fewer tiles re-reads its last one)
  if (mine) {
    const u4 *src = tile_src(0);
#pragma unroll
    for (int j = 0; j < CPT; j++) ld(R[0][j], src, j);
#pragma unroll
    for (int j = 0; j < CPT; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[0][j])::"memory");
#pragma unroll
    for (int j = 0; j < CPT; j++) quant_store(R[0][j], 0, j);
  }
#pragma unroll
  for (int d = 0; d < D; d++) {
    const u4 *src = tile_src((uint32_t)(1 + d) < mine ? 1 + d : (mine ? mine - 1 : 0));
#pragma unroll
    for (int j = 0; j < CPT; j++) ld(R[d][j], src, j);
  }
  // (drained once, here: hipcc may MOVE the ring's registers between this prologue and the loop -- it believes they hold their
  // values since the asm -- and a register copied while its load is in flight carries the old bits: tiles 1 .. D went wrong)
  // (no register operands: tying all D CPT registers to one statement cost 19 spills -- of in-flight registers -- at dim 768)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const uint32_t t16 = 16u * (h ^ (r & 15u));
  // one tile: matrix stream over buffer i & 1; the registers of set d (tile i + 1) are quantised into buffer (i + 1) & 1 and
  // re-requested (tile i + 1 + D) inside it
  auto step = [&](uint32_t i, u4 (&regs)[CPT]) {
    const uint32_t buf = i & 1u;
    const uint32_t tbase = lds_base + 16u * (buf * TILE + r * RS);
    const u4 *src = tile_src(i + 1 + D < mine ? i + 1 + D : i);
    typename AccT<KT_I8>::t acc[QB];
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
      for (int nb = 0; nb < QB; nb++) acc[nb][e] = 0;
    // (accumulators pinned to the accumulation registers: with everything in VGPRs hipcc let the first matrix instruction of a
    // tile write v[0:15] while reading its A fragment from v[0:3] -- rows went missing, differently from run to run)
#pragma unroll
    for (int nb = 0; nb < QB; nb++) asm volatile("" : "+a"(acc[nb]));
    constexpr int EVERY = KS / CPT > 0 ? KS / CPT : 1;  // k-steps between two chunks of the next tile
    auto frag = [&](int ks) {
      return *reinterpret_cast<const u4 *>(reinterpret_cast<const char *>(smem) + (tbase - lds_base) + 256u * (uint32_t)(ks >> 3) +
                                           ((32u * (uint32_t)(ks & 7)) ^ t16));
    };
    // fragments two k-steps ahead, the order pinned per k-step: left alone hipcc hoists a dozen ds_read_b128 to the top of the tile
    // (48 registers) -- with the fp32 form's twelve chunks per tile that spilled ring registers at dim 768
    constexpr int PF = 2;
    u4 fq[PF];
#pragma unroll
    for (int p = 0; p < PF; p++) fq[p] = frag(p < KS ? p : KS - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const u4 f = fq[ks % PF];
#pragma unroll
      for (int nb = 0; nb < QB; nb++) acc[nb] = mfma<KT_I8>(f, Q[nb][ks], acc[nb]);
      if (ks + PF < KS) fq[ks % PF] = frag(ks + PF);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < CPT; j++)
        if (j * EVERY == ks || (KS < CPT && ks == 0)) {
          // (unconditional: past the workgroup's last tile the store fills a buffer nobody reads and the load re-reads the
          // tile it just finished -- straight-line code keeps hipcc's vmcnt waits COUNTED; under `if (refill)` every wait of
          // the loop became vmcnt(0) and the tiles in flight were drained at every chunk)
          __builtin_amdgcn_sched_barrier(0);  // (pinned: hipcc hoists all CPT conversions to the top of the tile and drains vmcnt there)
          use(regs[j]);
          quant_store(regs[j], buf ^ 1u, j);
          ld(regs[j], src, j);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint32_t xr0 = row_first + i * row_step + 4 * h;
    bool stored = false;
#pragma unroll
    for (int nb = 0; nb < QB; nb++) {
      int m = max3i(acc[nb][0], acc[nb][1], acc[nb][2]);
#pragma unroll
      for (int e = 3; e < 15; e += 2) m = max3i(m, acc[nb][e], acc[nb][e + 1]);
      m = max3i(m, acc[nb][15], acc[nb][15]);
      if (m >= ithr[nb]) {
        uint2 *list = g.sub_cand + ((((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + r) * 2 + h) * g.sub_cap);
#pragma unroll
        for (int eg = 0; eg < 4; eg++) {
          const int gm = max3i(max3i(acc[nb][4 * eg], acc[nb][4 * eg + 1], acc[nb][4 * eg + 2]), acc[nb][4 * eg + 3], acc[nb][4 * eg + 3]);
          if (gm >= ithr[nb]) {
#pragma unroll
            for (int el = 0; el < 4; el++) {
              const float d = 1.0f - qsc[nb] * (float)acc[nb][4 * eg + el];
              const uint32_t xr = xr0 + 8 * eg + el;
              if (acc[nb][4 * eg + el] >= ithr[nb] && d <= tau[nb] && xr < row_end) {
                // (the store as inline asm: a store hipcc can see makes the vmcnt bracket "mixed" -- loads and stores pending -- and
                // every wait on the tiles in flight becomes a drain)
                if (cur[nb] < g.sub_cap) {
                  const uint2 *dst = list + cur[nb];
                  const uint2 val = make_uint2(xr, __float_as_uint(d));
                  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(val) : "memory");
                  stored = true;
                }
                cur[nb]++;
              }
            }
          }
        }
      }
    }
    // The refills of the last D tiles (re-reads of a finished tile) must land before the wave moves on: behind the loop hipcc
    // re-used v[2:3] -- dead to it, in flight in fact -- for the pointer of the final store, the load landed on it and the store
    // went to a wild address (MEMORY_APERTURE_VIOLATION); a wave that ENDS under its loads faults as well.  Drained here, in
    // the workgroup's last tile, where the ring is still live to the compiler (it cannot know this trip is the last).
    // A wave that stored candidates waits for them: vmcnt counts loads and stores alike, loads return in order among themselves
    // but a younger store may retire before an older load -- with stores in the queue the counted waits of the next tile could
    // pass before the register they guard has landed.  Drained, the counts are exact again; only the waves with a hit pay.
    if (i + 1 == mine || __builtin_amdgcn_ballot_w64(stored) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and the ring is LIVE to the compiler up to here, in every trip: the re-requests of the last D trips are never converted --
    // dead values to hipcc, which (peeling the last trip when one tile is in flight) parked them in registers it then used for
    // this epilogue's store pointers; the load landed on the pointer and the store went wild (MEMORY_APERTURE_VIOLATION).
#pragma unroll
    for (int j = 0; j < CPT; j++) asm volatile("" : "+v"(regs[j]));
    __syncthreads();  // buffer (i + 1) & 1 is complete, buffer i & 1 is free
  };
  for (uint32_t i = 0; i < mine; i += D) {
#pragma unroll
    for (int d = 0; d < D; d++)
      if (i + d < mine) step(i + d, R[d]);
  }
#pragma unroll
  for (int nb = 0; nb < QB; nb++)
    g.sub_count[((size_t)blockIdx.x * 256 + 32 * QB * w + 32 * nb + r) * 2 + h] = cur[nb];
}

// queries (fp32 rows, qstride bytes apart) -> bf16 rows of 2 * dim bytes padded to 16 (round to nearest even, the
// conversion the pass applies to the corpus): one wavefront per query
__global__ __launch_bounds__(256) void convert_queries_bf16_kernel(const float *__restrict__ q, uint32_t qstride4, uint32_t dim,
                                                                   uint32_t n_queries, uint32_t *__restrict__ out, uint32_t ostride4) {
  const uint32_t lane = threadIdx.x & 63, qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= n_queries) return;
  for (uint32_t c = lane; c < ostride4; c += 64) {
    const float a = 2 * c < dim ? q[(size_t)qi * qstride4 + 2 * c] : 0.0f, b = 2 * c + 1 < dim ? q[(size_t)qi * qstride4 + 2 * c + 1] : 0.0f;
    out[(size_t)qi * ostride4 + c] = cvt_pk_bf16(__float_as_uint(a), __float_as_uint(b));
  }
}

// One workgroup per query: concatenate its 2 * n_wg sub-lists (row, distance bits) into
// cand[q*cand_cap ..] as (row, orderable key) and set cand_count[q] (cand_cap+1 when a sub-list or the
// list itself overflowed -> the select flags the query and the host redoes it).
__global__ __launch_bounds__(1024) void compact_cand_kernel(const uint32_t *__restrict__ sub_count,
                                                           const uint2 *__restrict__ sub_cand, uint32_t sub_cap,
                                                           uint32_t n_wg, uint32_t *__restrict__ cand_count,
                                                           uint2 *__restrict__ cand, uint32_t cand_cap, int append,
                                                           RowBand band) {
  __shared__ uint32_t offs[513];
  __shared__ uint32_t wsum[8];
  __shared__ uint32_t over;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) over = 0;
  __syncthreads();
  // the sub-lists' starts: a prefix over 512 counts by eight wavefronts (round 6: one thread summed them through LDS, and a
  // wavefront copied its 32 sub-lists one dependent round trip after the other -- 30 us per launch, four to five launches per pass)
  uint32_t c = 0;
  if (tid < 512) {  // segment sg = (workgroup sg/2, lane half sg%2)
    c = tid < 2 * n_wg ? sub_count[((size_t)(tid >> 1) * 256 + q) * 2 + (tid & 1)] : 0;
    if (c > sub_cap) {
      over = 1;
      c = sub_cap;
    }
  }
  uint32_t inc = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc += t;
  }
  if (tid < 512 && lane == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  if (tid < 512) {
    uint32_t add = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) add += wsum[w];
    offs[tid + 1] = inc + add;
  }
  if (tid == 0) offs[0] = 0;
  __syncthreads();
  const uint32_t base = append ? cand_count[q] : 0;  // (every thread reads it before thread 0 rewrites it below)
  const uint32_t total = offs[512];
  __syncthreads();
  if (base > cand_cap || total > cand_cap - base) {
    if (tid == 0) cand_count[q] = cand_cap + 1;
    return;
  }
  // entry o of the concatenation: its sub-list by a search over the starts, every load independent of every other
  for (uint32_t o = tid; o < total; o += blockDim.x) {
    uint32_t lo = 0, hi = 512;  // offs[lo] <= o < offs[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (offs[mid] <= o) lo = mid;
      else hi = mid;
    }
    const uint32_t sg = lo, i = o - offs[sg];
    const uint2 e = sub_cand[(((size_t)(sg >> 1) * 256 + q) * 2 + (sg & 1)) * sub_cap + i];
    float d = __uint_as_float(e.y);
    if (band.hnorm) d += band.c1 * band.hnorm[e.x] + band.hq2[q];  // lower bound -> upper bound (L2 passes)
    cand[(size_t)q * cand_cap + base + o] = make_uint2(e.x, f2key(d));
  }
  if (tid == 0) cand_count[q] = over ? cand_cap + 1 : base + total;
}

template <int DT, int KS, int NS>
void launch_qs_shape(const QsArgs &g, uint32_t grid, hipStream_t s) {
  if constexpr (DT != KT_I8) {
    if (g.hnorm) {  // L2 (eight waves x 32 queries only)
      hipLaunchKernelGGL((gemm_qs_kernel<DT, KS, NS, 1, true>), dim3(grid), dim3(512), 0, s, g);
      return;
    }
  }
  switch (scan_tuning().gemm_qs) {  // 1: eight waves x 32 queries (default); 2: four waves x 64 queries
    case 2: hipLaunchKernelGGL((gemm_qs_kernel<DT, KS, NS, 2, false>), dim3(grid), dim3(256), 0, s, g); break;
    default:
      hipLaunchKernelGGL((gemm_qs_kernel<DT, KS, NS, 1, false>), dim3(grid), dim3(512), 0, s, g);
      break;
  }
}

template <int DT>
bool launch_qs_dt(const QsArgs &g, uint32_t stride16, uint32_t grid, hipStream_t s) {
  switch (stride16) {  // chunks per row = 2 * KS; the ring is as deep as ~144 KiB of LDS allows
    case 96:
      // (int8 rows of 1536 bytes are not offered: untested shape)
      if constexpr (DT == KT_I8) return false;
      else {
        launch_qs_shape<DT, 48, 3>(g, grid, s);
        return true;
      }
    case 64: launch_qs_shape<DT, 32, 4>(g, grid, s); return true;
    case 48: launch_qs_shape<DT, 24, 6>(g, grid, s); return true;
    case 32: launch_qs_shape<DT, 16, 8>(g, grid, s); return true;
    case 16: launch_qs_shape<DT, 8, 8>(g, grid, s); return true;
    default: return false;
  }
}

}  // namespace

bool gemm_qs_supported(uint32_t stride16) {
  return stride16 == 96 || stride16 == 64 || stride16 == 48 || stride16 == 32 || stride16 == 16;
}

uint32_t gemm_qs_grid(uint32_t n_rows) {
  const uint32_t cus = (uint32_t)scan_tuning().num_cus, tiles = (n_rows + 31) / 32;
  const uint32_t cap = cus < 256 ? cus : 256;  // compact_cand_kernel handles up to 256 sub-lists
  return tiles < cap ? tiles : cap;
}

bool launch_gemm_qs(int dtype, const void *rows, const void *queries, uint32_t stride16, uint32_t row_begin,
                    uint32_t row_end, const float *tau, uint32_t *sub_count, void *sub_cand, uint32_t sub_cap,
                    hipStream_t s, const float *qscale, const float *hnorm, const float *hq) {
  if (row_end <= row_begin || !gemm_qs_supported(stride16)) return false;
  if ((hnorm != nullptr) != (hq != nullptr) || (hnorm && dtype == KT_I8)) return false;
  QsArgs g{(const u4 *)rows, (const u4 *)queries, row_begin, row_end, tau, sub_count, (uint2 *)sub_cand, sub_cap, qscale, hnorm, hq};
  const uint32_t grid = gemm_qs_grid(row_end - row_begin);
  if (dtype == KT_I8 || (scan_tuning().qs_force_i8 && !hnorm)) {  // (qs_force_i8: timing experiment over bytes that are not int8 data)
    if (!g.qscale) g.qscale = tau;
    return launch_qs_dt<KT_I8>(g, stride16, grid, s);
  }
  return dtype == KT_F16 ? launch_qs_dt<KT_F16>(g, stride16, grid, s) : launch_qs_dt<KT_BF16>(g, stride16, grid, s);
}

// ---- FLOAT32 rows (gemm_qs_f32_kernel): stride16 = fp32 chunks per row -----------------------------------------------------
bool gemm_qs_f32_supported(uint32_t stride16) {
  return stride16 == 192 || stride16 == 128 || stride16 == 96 || stride16 == 64 || stride16 == 32;
}

namespace {
template <int KS, int KH, int NS>
void launch_qs_f32_shape(const QsArgs &g, uint32_t grid, hipStream_t s) {
  const bool wide = scan_tuning().gemm_qs_f32 != 1;  // default: four waves x 64 queries (one 512-register wave per SIMD); 1: eight x 32
  if (g.hnorm) {
    if (wide) hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KH, NS, 2, true>), dim3(grid), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KH, NS, 1, true>), dim3(grid), dim3(512), 0, s, g);
  } else {
    if (wide) hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KH, NS, 2, false>), dim3(grid), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KH, NS, 1, false>), dim3(grid), dim3(512), 0, s, g);
  }
}
}  // namespace

bool launch_gemm_qs_f32(const void *rows, const void *queries_bf16, uint32_t stride16, uint32_t row_begin, uint32_t row_end,
                        const float *tau, uint32_t *sub_count, void *sub_cand, uint32_t sub_cap, hipStream_t s, const float *hnorm,
                        const float *hq) {
  if (row_end <= row_begin || !gemm_qs_f32_supported(stride16) || (hnorm != nullptr) != (hq != nullptr)) return false;
  QsArgs g{(const u4 *)rows, (const u4 *)queries_bf16, row_begin, row_end, tau, sub_count, (uint2 *)sub_cand, sub_cap, nullptr, hnorm, hq};
  const uint32_t grid = gemm_qs_grid(row_end - row_begin);
  switch (stride16) {  // KS = dim / 16 k-steps; slots of 2 KiB x KH, the ring as deep as ~144 KiB of LDS allows
    case 192: launch_qs_f32_shape<48, 24, 3>(g, grid, s); return true;
    case 128: launch_qs_f32_shape<32, 16, 4>(g, grid, s); return true;
    case 96: launch_qs_f32_shape<24, 24, 3>(g, grid, s); return true;
    case 64: launch_qs_f32_shape<16, 16, 4>(g, grid, s); return true;
    case 32: launch_qs_f32_shape<8, 8, 8>(g, grid, s); return true;
    default: return false;
  }
}

// ---- FLOAT16 rows quantised to int8 in flight (SRC_H8): stride16 = fp16 chunks per row, queries = int8 rows of stride16 * 8 bytes
bool gemm_qs_h8_supported(uint32_t stride16) { return stride16 == 96 || stride16 == 64 || stride16 == 48 || stride16 == 32 || stride16 == 16; }

namespace {
template <int KS, int NS>
void launch_qs_h8_shape(const QsArgs &g, uint32_t grid, hipStream_t s) {
  // 2 (default): four waves x 64 queries -- every quantised fragment feeds two matrix instructions; 1: eight waves x 32
  // 5: the register-staged form (gemm_qs_h8r_kernel: quantised once per workgroup), eight waves x 32 queries, two tiles in flight
  // (four waves x 64 queries measured 7.8 ms per pass -- one wave per SIMD hides nothing here; three tiles in flight spill at dim 768)
  if (scan_tuning().gemm_qs_h8 == 5) hipLaunchKernelGGL((gemm_qs_h8r_kernel<KS, 1, 2>), dim3(grid), dim3(512), 0, s, g);
  else if (scan_tuning().gemm_qs_h8 == 8) hipLaunchKernelGGL((gemm_qs_h8r_kernel<KS, 1, 1>), dim3(grid), dim3(512), 0, s, g);  // (A/B: one tile in flight)
  else if (scan_tuning().gemm_qs_h8 == 1) hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KS, NS, 1, false, SRC_H8>), dim3(grid), dim3(512), 0, s, g);
  else if (KS == 24 && scan_tuning().gemm_qs_h8 == 3)  // (A/B: K-part slots of 24 KiB, a ring of six)
    hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KS == 24 ? 12 : KS, KS == 24 ? 6 : NS, 2, false, SRC_H8>), dim3(grid), dim3(256), 0, s, g);
  else if (KS == 24 && scan_tuning().gemm_qs_h8 == 4)  // (A/B: slots of 16 KiB, a ring of nine)
    hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KS == 24 ? 8 : KS, KS == 24 ? 9 : NS, 2, false, SRC_H8>), dim3(grid), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_qs_f32_kernel<KS, KS, NS, 2, false, SRC_H8>), dim3(grid), dim3(256), 0, s, g);
}
}  // namespace

bool launch_gemm_qs_h8(const void *rows, const void *queries_i8, uint32_t stride16, uint32_t row_begin, uint32_t row_end, const float *tau,
                       uint32_t *sub_count, void *sub_cand, uint32_t sub_cap, hipStream_t s, const float *qscale, uint16_t inv_h_bits) {
  if (row_end <= row_begin || !gemm_qs_h8_supported(stride16) || !qscale) return false;
  QsArgs g{(const u4 *)rows, (const u4 *)queries_i8, row_begin, row_end, tau, sub_count, (uint2 *)sub_cand, sub_cap, qscale, nullptr, nullptr,
           (uint32_t)inv_h_bits | ((uint32_t)inv_h_bits << 16)};
  const uint32_t grid = gemm_qs_grid(row_end - row_begin);
  switch (stride16) {  // KS = dim / 32 int8 k-steps = stride16 / 4; slots of 32 rows x 64 B x KS, the ring as deep as ~144 KiB allow
    case 96: launch_qs_h8_shape<24, 3>(g, grid, s); return true;
    case 64: launch_qs_h8_shape<16, 4>(g, grid, s); return true;
    case 48: launch_qs_h8_shape<12, 6>(g, grid, s); return true;
    case 32: launch_qs_h8_shape<8, 8>(g, grid, s); return true;
    case 16: launch_qs_h8_shape<4, 8>(g, grid, s); return true;
    default: return false;
  }
}

// ---- FLOAT32 rows quantised to int8 once per workgroup (gemm_qs_h8r_kernel<.., SRC_F8>): stride16 = fp32 chunks per row
bool gemm_qs_f8_supported(uint32_t stride16) { return stride16 == 192 || stride16 == 128 || stride16 == 96 || stride16 == 64 || stride16 == 32; }
bool launch_gemm_qs_f8(const void *rows, const void *queries_i8, uint32_t stride16, uint32_t row_begin, uint32_t row_end, const float *tau,
                       uint32_t *sub_count, void *sub_cand, uint32_t sub_cap, hipStream_t s, const float *qscale, float inv) {
  if (row_end <= row_begin || !gemm_qs_f8_supported(stride16) || !qscale) return false;
  uint32_t inv_bits;
  memcpy(&inv_bits, &inv, 4);
  QsArgs g{(const u4 *)rows, (const u4 *)queries_i8, row_begin, row_end, tau, sub_count, (uint2 *)sub_cand, sub_cap, qscale, nullptr, nullptr, inv_bits};
  const uint32_t grid = gemm_qs_grid(row_end - row_begin);
  // (ONE tile in flight per thread: an fp32 tile is 96 KiB -- what two fp16 tiles are -- and two of them do not fit 256 registers)
  switch (stride16) {  // KS = dim / 32 = stride16 / 8
    case 192: hipLaunchKernelGGL((gemm_qs_h8r_kernel<24, 1, 1, SRC_F8>), dim3(grid), dim3(512), 0, s, g); return true;
    case 128: hipLaunchKernelGGL((gemm_qs_h8r_kernel<16, 1, 1, SRC_F8>), dim3(grid), dim3(512), 0, s, g); return true;
    case 96: hipLaunchKernelGGL((gemm_qs_h8r_kernel<12, 1, 1, SRC_F8>), dim3(grid), dim3(512), 0, s, g); return true;
    case 64: hipLaunchKernelGGL((gemm_qs_h8r_kernel<8, 1, 1, SRC_F8>), dim3(grid), dim3(512), 0, s, g); return true;
    case 32: hipLaunchKernelGGL((gemm_qs_h8r_kernel<4, 1, 1, SRC_F8>), dim3(grid), dim3(512), 0, s, g); return true;
    default: return false;
  }
}

void launch_convert_queries_bf16(const void *queries, size_t qstride, uint32_t dim, uint32_t n_queries, void *out, size_t ostride,
                                 hipStream_t s) {
  if (!n_queries) return;
  hipLaunchKernelGGL(convert_queries_bf16_kernel, dim3((n_queries + 3) / 4), dim3(256), 0, s, (const float *)queries,
                     (uint32_t)(qstride / 4), dim, n_queries, (uint32_t *)out, (uint32_t)(ostride / 4));
}

void launch_compact_cand(const uint32_t *sub_count, const void *sub_cand, uint32_t sub_cap, uint32_t n_wg,
                         uint32_t *cand_count, void *cand, uint32_t cand_cap, int append, hipStream_t s, const RowBand *band) {
  hipLaunchKernelGGL(compact_cand_kernel, dim3(256), dim3(1024), 0, s, sub_count, (const uint2 *)sub_cand, sub_cap, n_wg,
                     cand_count, (uint2 *)cand, cand_cap, append, band ? *band : RowBand{});
}

}  // namespace rsgpu
