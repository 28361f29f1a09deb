// select_kernels.hip -- GPU-side exact top-K selection and range compaction over the key array the
// scan wrote (gfx950, hand-written HIP).
//
// Replaces the K-bounded heap inside VecSimIndex_TopKQuery / VecSimBatchIterator_Next and the
// iterator-side heap bookkeeping (reference src/iterators/hybrid_reader.c:88-138,372-443).
//
// Order.  Every row has the unique 64-bit composite  (key << 32) | row, key being the orderable
// image of its fp32 distance.  "Top K" = the K smallest composites, i.e. ascending distance with
// ties resolved by storage row -- the order in which the reference's scan would have met them
// (oracle/flat_oracle.c header, [upstream-memory D3]).  Because the order is total the selection
// is exact and deterministic for any K, any number of equal distances, and resumable: a batch
// iterator restarts above the previous batch's largest composite (`lower`).
//
// Algorithm.  MSB-first radix select, 8 bits per level.  Level p histograms digit p of the
// candidates that match the digits chosen at levels 0..p-1; the histograms of all levels stay in
// global memory, and every workgroup of the next kernel re-derives the chosen prefix from them
// (256 counters per level -- a few hundred cycles) instead of waiting on an inter-workgroup
// hand-off: kernel boundaries are the only synchronisation, so nothing depends on dispatch order or
// XCD placement.  Selection is exact as soon as the chosen bucket holds exactly the number of
// elements still wanted; for continuous data that happens after the 2nd-4th level, and levels 4-7
// (row bits) run only when equal keys straddle rank K.
//
// Cost: each level is one coalesced pass over 4 B/row (40 MB at 10 M rows, ~8 us at HBM speed) --
// about 1 % of the 30.7 GB scan it follows.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace rsgpu {
namespace {

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

struct Decision {
  unsigned long long prefix;  // chosen digits, left-aligned
  uint32_t k_rem;             // still wanted inside the chosen bucket
  int levels;                 // digits chosen (prefix valid in the top 8*levels bits)
  int exact;                  // chosen bucket count == k_rem (or fewer candidates than k)
  int take_all;               // fewer candidates than k: everything above `lower` is selected
};

// Re-derive the decision from the histograms of levels [0, passes). Executed by wavefront 0 of every
// workgroup, result published through LDS.
__device__ void decide(const uint32_t *__restrict__ hist, int passes, uint32_t k, Decision *out) {
  if (threadIdx.x < 64) {
    const uint32_t lane = threadIdx.x;
    unsigned long long prefix = 0;
    uint32_t k_rem = k;
    int exact = 0, levels = 0, take_all = 0;
    for (int p = 0; p < passes && !exact; ++p) {
      const uint32_t *h = hist + p * 256 + lane * 4;
      uint32_t h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
      uint32_t s = h0 + h1 + h2 + h3;
      uint32_t inc = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += t;
      }
      uint32_t exc = inc - s;
      bool mine = (exc < k_rem) && (k_rem <= inc);
      unsigned long long ball = __ballot(mine);
      if (ball == 0) {  // fewer candidates than k (host clamps k, so only on an empty remainder)
        exact = 1;
        take_all = 1;
        break;
      }
      int src = __ffsll((long long)ball) - 1;
      uint32_t D = 0, below = 0, cnt = 0;
      if (lane == (uint32_t)src) {
        uint32_t c = exc;
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
      }
      D = __shfl(D, src, 64);
      below = __shfl(below, src, 64);
      cnt = __shfl(cnt, src, 64);
      k_rem -= below;
      prefix |= (unsigned long long)D << (56 - 8 * p);
      levels = p + 1;
      if (cnt == k_rem) exact = 1;
    }
    if (lane == 0) {
      out->prefix = prefix;
      out->k_rem = k_rem;
      out->levels = levels;
      out->exact = exact;
      out->take_all = take_all;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ unsigned long long comp_of(uint32_t key, uint32_t row) {
  return ((unsigned long long)key << 32) | row;
}

// ---- one histogram level ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_pass_kernel(const uint32_t *__restrict__ keys, uint32_t n, int pass,
                                                          uint32_t k, unsigned long long lower, int has_lower,
                                                          uint32_t *__restrict__ hist) {
  __shared__ Decision dec;
  __shared__ uint32_t lh[256];
  lh[threadIdx.x] = 0;
  decide(hist, pass, k, &dec);  // ends with __syncthreads()
  if (dec.exact) return;        // already resolved at an earlier level
  const unsigned long long prefix = dec.prefix;
  const int mshift = 64 - 8 * pass;  // bits below the matched prefix (64 at pass 0: no prefix)
  const int dshift = 56 - 8 * pass;
  const uint32_t lane = threadIdx.x & 63;

  const uint32_t n4 = (n + 3) / 4;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ((n4 + 255) / 256) * 256; i += gridDim.x * 256) {
    u4 kv = (u4){0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const uint32_t base = i * 4;
    if (i < n4) {
      if (base + 3 < n) kv = *(const u4 *)(keys + base);
      else {
        kv.x = keys[base];
        if (base + 1 < n) kv.y = keys[base + 1];
        if (base + 2 < n) kv.z = keys[base + 2];
      }
    }
    uint32_t kk[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t row = base + j;
      const unsigned long long c = comp_of(kk[j], row);
      bool active = (i < n4) && (row < n);
      if (has_lower) active = active && (c > lower);
      if (pass > 0) active = active && ((c >> mshift) == (prefix >> mshift));
      const uint32_t d = (uint32_t)(c >> dshift) & 0xffu;
      // wave-aggregated LDS increment when the whole wavefront agrees on the digit (the common case
      // at the top levels: distances cluster in a handful of exponent buckets)
      unsigned long long act = __ballot(active);
      if (act) {
        int first = __ffsll((long long)act) - 1;
        uint32_t d0 = __shfl(d, first, 64);
        unsigned long long same = __ballot(active && d == d0);
        if (same == act) {
          if (lane == (uint32_t)first) atomicAdd(&lh[d0], (uint32_t)__popcll(act));
        } else if (active) {
          atomicAdd(&lh[d], 1u);
        }
      }
    }
  }
  __syncthreads();
  uint32_t v = lh[threadIdx.x];
  if (v) atomicAdd(&hist[pass * 256 + threadIdx.x], v);
}

// append one element per active lane with a single atomic per wavefront
__device__ __forceinline__ void wave_append(bool take, uint32_t row, uint32_t key, uint32_t *cursor,
                                            uint32_t *out_rows, uint32_t *out_keys, uint32_t cap) {
  unsigned long long m = __ballot(take);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63;
  int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if (lane == (uint32_t)leader) base = atomicAdd(cursor, (uint32_t)__popcll(m));
  base = __shfl(base, leader, 64);
  if (take) {
    uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot < cap) {
      out_rows[slot] = row;
      out_keys[slot] = key;
    }
  }
}

__global__ __launch_bounds__(256) void select_collect_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                             int passes_done, uint32_t k,
                                                             unsigned long long lower, int has_lower,
                                                             const uint32_t *__restrict__ hist,
                                                             uint32_t *__restrict__ counters,
                                                             uint32_t *__restrict__ out_rows,
                                                             uint32_t *__restrict__ out_keys,
                                                             unsigned long long *__restrict__ bound, uint32_t cap) {
  __shared__ Decision dec;
  decide(hist, passes_done, k, &dec);
  if (!dec.exact) {
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[1] = 1;  // caller runs more levels
    return;
  }
  const int shift = 64 - 8 * dec.levels;  // unrefined low bits
  const unsigned long long hi =
      dec.take_all ? ~0ull : (shift >= 64 ? ~0ull : (dec.prefix | ((shift > 0) ? ((1ull << shift) - 1ull) : 0ull)));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counters[1] = 0;
    bound[0] = hi;
  }
  const uint32_t n4 = (n + 3) / 4;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ((n4 + 255) / 256) * 256; i += gridDim.x * 256) {
    u4 kv = (u4){0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const uint32_t base = i * 4;
    if (i < n4) {
      if (base + 3 < n) kv = *(const u4 *)(keys + base);
      else {
        kv.x = keys[base];
        if (base + 1 < n) kv.y = keys[base + 1];
        if (base + 2 < n) kv.z = keys[base + 2];
      }
    }
    uint32_t kk[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t row = base + j;
      const unsigned long long c = comp_of(kk[j], row);
      bool take = (i < n4) && (row < n) && (c <= hi);
      if (has_lower) take = take && (c > lower);
      wave_append(take, row, kk[j], &counters[0], out_rows, out_keys, cap);
    }
  }
}

// ---- range ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void range_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t max_key,
                                                    int collect, uint32_t *__restrict__ counters,
                                                    uint32_t *__restrict__ out_rows, uint32_t *__restrict__ out_keys,
                                                    uint32_t cap) {
  uint32_t local = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ((n + 255) / 256) * 256; i += gridDim.x * 256) {
    uint32_t key = i < n ? keys[i] : 0xFFFFFFFFu;
    bool take = i < n && key <= max_key;
    if (collect) wave_append(take, i, key, &counters[0], out_rows, out_keys, cap);
    else local += take ? 1u : 0u;
  }
  if (!collect) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) local += __shfl_xor(local, m, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counters[0], local);
  }
}

inline uint32_t pass_grid(uint32_t n) {
  uint32_t need = ((n + 3) / 4 + 255) / 256;
  uint32_t cap = (uint32_t)scan_tuning().num_cus * 2;
  uint32_t g = need < cap ? need : cap;
  return g ? g : 1;
}

}  // namespace

void launch_select_pass(const uint32_t *keys, uint32_t n, int pass, uint32_t k, uint64_t lower, int has_lower,
                        const SelectBufs &b, hipStream_t s) {
  hipLaunchKernelGGL(select_pass_kernel, dim3(pass_grid(n)), dim3(256), 0, s, keys, n, pass, k,
                     (unsigned long long)lower, has_lower, b.hist);
}

void launch_select_collect(const uint32_t *keys, uint32_t n, int passes_done, uint32_t k, uint64_t lower,
                           int has_lower, const SelectBufs &b, uint32_t cap, hipStream_t s) {
  hipLaunchKernelGGL(select_collect_kernel, dim3(pass_grid(n)), dim3(256), 0, s, keys, n, passes_done, k,
                     (unsigned long long)lower, has_lower, b.hist, b.counters, b.out_rows, b.out_keys,
                     (unsigned long long *)b.bound, cap);
}

void launch_range(const uint32_t *keys, uint32_t n, uint32_t max_key, int collect, uint32_t *counters,
                  uint32_t *out_rows, uint32_t *out_keys, uint32_t cap, hipStream_t s) {
  uint32_t need = (n + 255) / 256, cap_g = (uint32_t)scan_tuning().num_cus * 4;
  uint32_t g = need < cap_g ? need : cap_g;
  hipLaunchKernelGGL(range_kernel, dim3(g ? g : 1), dim3(256), 0, s, keys, n, max_key, collect, counters, out_rows,
                     out_keys, cap);
}

}  // namespace rsgpu
