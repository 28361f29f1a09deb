// select_kernels.hip -- GPU-side exact top-K selection and range compaction over a key array
// (gfx950, hand-written HIP).
//
// Replaces the K-bounded heap inside VecSimIndex_TopKQuery / VecSimBatchIterator_Next, the
// iterator-side heap bookkeeping (reference src/iterators/hybrid_reader.c:88-138,372-443) and, with
// 64-bit keys, the result sorter's top-N heap over scores (reference src/result_processor.c:752-850).
//
// Order.  Element i has the unique composite (key_i, i): key is the orderable image of an fp32
// distance (u32) or of an fp64 score (u64), i is the storage row / hit index.  "Top K" = the K
// smallest composites, i.e. ascending key with ties resolved by position -- the order in which the
// reference's sequential scan meets the elements (oracle/flat_oracle.c header, [upstream-memory D3];
// reference src/result_processor.c:849 for scores: equal score => lower doc id first).  The order is
// total, so the selection is exact and deterministic for any K and any number of equal keys, and it
// is resumable: a batch iterator restarts above the previous batch's largest composite (`lower`).
//
// Algorithm.  MSB-first radix select, 8 bits per level, sizeof(key)+4 levels at most.  Level p
// histograms digit p of the candidates that match the digits chosen at levels 0..p-1; all levels'
// histograms stay in global memory and every workgroup of the next kernel re-derives the chosen
// prefix from them (256 counters per level -- a few hundred cycles) instead of waiting on an
// inter-workgroup hand-off: kernel boundaries are the only synchronisation, nothing depends on
// dispatch order or XCD placement.  Selection is exact as soon as the chosen bucket holds exactly the
// number of elements still wanted; the position levels run only when equal keys straddle rank K.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace rsgpu {
namespace {

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

struct Decision {
  u64 pkey;        // chosen key digits, left-aligned inside the key width
  uint32_t prow;   // chosen row digits, left-aligned
  uint32_t k_rem;  // still wanted inside the chosen bucket
  int levels;      // digits chosen so far
  int exact;       // chosen bucket count == k_rem (or fewer candidates than k)
  int take_all;    // fewer candidates than k: everything above `lower` is selected
};

// Re-derive the decision from the histograms of levels [0, passes). Executed by wavefront 0 of every
// workgroup, published through LDS.  KB = key bytes.
template <int KB>
__device__ void decide(const uint32_t *__restrict__ hist, int passes, uint32_t k, Decision *out) {
  if (threadIdx.x < 64) {
    const uint32_t lane = threadIdx.x;
    u64 pkey = 0;
    uint32_t prow = 0, k_rem = k;
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
      u64 ball = __ballot(mine);
      if (ball == 0) {  // fewer candidates than k (the host clamps k; only an empty remainder gets here)
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
      if (p < KB) pkey |= (u64)D << (8 * (KB - 1 - p));
      else prow |= D << (8 * (3 - (p - KB)));
      levels = p + 1;
      if (cnt == k_rem) exact = 1;
    }
    if (lane == 0) {
      out->pkey = pkey;
      out->prow = prow;
      out->k_rem = k_rem;
      out->levels = levels;
      out->exact = exact;
      out->take_all = take_all;
    }
  }
  __syncthreads();
}

template <int KB>
__device__ __forceinline__ uint32_t digit_of(int p, u64 key, uint32_t row) {
  return p < KB ? (uint32_t)(key >> (8 * (KB - 1 - p))) & 0xffu : (row >> (8 * (3 - (p - KB)))) & 0xffu;
}
// do the top p digits of (key,row) equal those of (pkey,prow)?
template <int KB>
__device__ __forceinline__ bool prefix_match(int p, u64 key, uint32_t row, u64 pkey, uint32_t prow) {
  if (p == 0) return true;
  if (p <= KB) {
    int sh = 8 * (KB - p);
    return (key >> sh) == (pkey >> sh);
  }
  int sh = 8 * (4 - (p - KB));
  return key == pkey && (sh >= 32 ? true : (row >> sh) == (prow >> sh));
}
__device__ __forceinline__ bool comp_gt(u64 key, uint32_t row, u64 lkey, uint32_t lrow) {
  return key > lkey || (key == lkey && row > lrow);
}
__device__ __forceinline__ bool comp_le(u64 key, uint32_t row, u64 hkey, uint32_t hrow) {
  return key < hkey || (key == hkey && row <= hrow);
}

// 4 consecutive keys starting at element 4*i (tail-safe); inactive slots read as all-ones
template <typename KeyT>
__device__ __forceinline__ void load4(const KeyT *__restrict__ keys, uint32_t n, uint32_t i, bool in_range, u64 out[4]) {
  const uint32_t base = i * 4;
#pragma unroll
  for (int j = 0; j < 4; j++) out[j] = ~0ull;
  if (!in_range) return;
  if (base + 3 < n) {
    if (sizeof(KeyT) == 4) {
      u4 v = *(const u4 *)(keys + base);
      out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
      const u4 *p = (const u4 *)(keys + base);
      u4 a = p[0], b = p[1];
      out[0] = (u64)a.x | ((u64)a.y << 32); out[1] = (u64)a.z | ((u64)a.w << 32);
      out[2] = (u64)b.x | ((u64)b.y << 32); out[3] = (u64)b.z | ((u64)b.w << 32);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (base + j < n) out[j] = (u64)keys[base + j];
  }
}

// ---- one histogram level ---------------------------------------------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(256) void select_pass_kernel(const KeyT *__restrict__ keys, uint32_t n, int pass,
                                                          uint32_t k, u64 lkey, uint32_t lrow, int has_lower,
                                                          uint32_t *__restrict__ hist) {
  constexpr int KB = sizeof(KeyT);
  __shared__ Decision dec;
  __shared__ uint32_t lh[256];
  lh[threadIdx.x] = 0;
  decide<KB>(hist, pass, k, &dec);  // ends with __syncthreads()
  if (dec.exact) return;            // already resolved at an earlier level
  const u64 pkey = dec.pkey;
  const uint32_t prow = dec.prow;
  const uint32_t lane = threadIdx.x & 63;

  const uint32_t n4 = (n + 3) / 4;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ((n4 + 255) / 256) * 256; i += gridDim.x * 256) {
    u64 kk[4];
    load4<KeyT>(keys, n, i, i < n4, kk);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t row = i * 4 + j;
      bool active = (i < n4) && (row < n);
      if (has_lower) active = active && comp_gt(kk[j], row, lkey, lrow);
      active = active && prefix_match<KB>(pass, kk[j], row, pkey, prow);
      const uint32_t d = digit_of<KB>(pass, kk[j], row);
      // wave-aggregated LDS increment when the whole wavefront agrees on the digit (the common case
      // at the top levels: distances cluster in a handful of exponent buckets)
      u64 act = __ballot(active);
      if (act) {
        int first = __ffsll((long long)act) - 1;
        uint32_t d0 = __shfl(d, first, 64);
        u64 same = __ballot(active && d == d0);
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
template <typename KeyT>
__device__ __forceinline__ void wave_append(bool take, uint32_t row, KeyT key, uint32_t *cursor,
                                            uint32_t *out_rows, KeyT *out_keys, uint32_t cap) {
  u64 m = __ballot(take);
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

template <typename KeyT>
__global__ __launch_bounds__(256) void select_collect_kernel(const KeyT *__restrict__ keys, uint32_t n,
                                                             int passes_done, uint32_t k, u64 lkey, uint32_t lrow,
                                                             int has_lower, const uint32_t *__restrict__ hist,
                                                             uint32_t *__restrict__ counters,
                                                             uint32_t *__restrict__ out_rows,
                                                             KeyT *__restrict__ out_keys, u64 *__restrict__ bound,
                                                             uint32_t cap) {
  constexpr int KB = sizeof(KeyT);
  __shared__ Decision dec;
  decide<KB>(hist, passes_done, k, &dec);
  if (!dec.exact) {
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[1] = 1;  // caller runs more levels
    return;
  }
  // inclusive upper bound of the selected set: chosen prefix, unrefined digits all ones
  u64 hkey = ~0ull;
  uint32_t hrow = 0xFFFFFFFFu;
  if (!dec.take_all) {
    const int L = dec.levels;
    if (L <= KB) {
      int sh = 8 * (KB - L);
      hkey = dec.pkey | (sh > 0 ? ((1ull << sh) - 1ull) : 0ull);
    } else {
      int sh = 8 * (4 - (L - KB));
      hkey = dec.pkey;
      hrow = dec.prow | (sh > 0 ? ((1u << sh) - 1u) : 0u);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counters[1] = 0;
    bound[0] = hkey;
    bound[1] = hrow;
  }
  const uint32_t n4 = (n + 3) / 4;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ((n4 + 255) / 256) * 256; i += gridDim.x * 256) {
    u64 kk[4];
    load4<KeyT>(keys, n, i, i < n4, kk);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t row = i * 4 + j;
      bool take = (i < n4) && (row < n) && comp_le(kk[j], row, hkey, hrow);
      if (has_lower) take = take && comp_gt(kk[j], row, lkey, lrow);
      wave_append<KeyT>(take, row, (KeyT)kk[j], &counters[0], out_rows, out_keys, cap);
    }
  }
}

// ---- range ---------------------------------------------------------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(256) void range_kernel(const KeyT *__restrict__ keys, uint32_t n, KeyT max_key,
                                                    int collect, uint32_t *__restrict__ counters,
                                                    uint32_t *__restrict__ out_rows, KeyT *__restrict__ out_keys,
                                                    uint32_t cap) {
  uint32_t local = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ((n + 255) / 256) * 256; i += gridDim.x * 256) {
    KeyT key = i < n ? keys[i] : (KeyT)~(KeyT)0;
    bool take = i < n && key <= max_key;
    if (collect) wave_append<KeyT>(take, i, key, &counters[0], out_rows, out_keys, cap);
    else local += take ? 1u : 0u;
  }
  if (!collect) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) local += __shfl_xor(local, m, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counters[0], local);
  }
}

// ---- threshold filter --------------------------------------------------------------------------------
// The fast top-K path for small K: a per-query upper bound tau of the K-th distance (K-th smallest of a
// strided sample, gemm_kernels.hip batch_select_kernel) turns the selection into ONE streaming pass that
// keeps the few keys <= tau, followed by a single-workgroup exact select of those candidates -- instead
// of four histogram passes over all keys.  tau is a true upper bound, so the result is exact; only an
// adversarial order can overflow the candidate buffer, and then the radix path above takes over.
__device__ __forceinline__ uint32_t f2key_dev(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xFFFFFFFFu;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// tau = K-th smallest of 1024 group minima over a sample of `per` keys per group.  The sample is per/4
// slabs of 4096 consecutive keys spread evenly over the array (robust to sorted corpora); thread t owns
// the t-th 16-byte chunk of every slab, so each load instruction is a fully coalesced 16 KiB sweep and all
// of a thread's loads are independent.  K distinct keys are <= tau, so tau bounds the K-th smallest key of
// the whole array.  One workgroup.
// Several queries at once (a coalesced pass, scan_mq_kernels.hip): grid (1, B), query b reads keys + b * keys_ld, writes
// tau_out[b] and zeroes zero_q[b] (its candidate counter).
__global__ __launch_bounds__(1024) void sample_threshold_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                                uint32_t slab_stride, uint32_t slabs, uint32_t k,
                                                                float *__restrict__ tau_out,
                                                                uint32_t *__restrict__ zero4, uint32_t keys_ld,
                                                                uint32_t *__restrict__ zero_q) {
  const uint32_t t = threadIdx.x;
  keys += (size_t)blockIdx.y * keys_ld;
  tau_out += blockIdx.y;
  if (zero4 && t < 4) zero4[t] = 0;  // the filter pass's counters: saves a memset on the query's critical path
  if (zero_q && t == 0) zero_q[blockIdx.y] = 0;
  uint32_t m = 0xFFFFFFFFu;
  for (uint32_t c0 = 0; c0 < slabs; c0 += 16) {
    u4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const uint32_t c = c0 + u < slabs ? c0 + u : slabs - 1;  // (repeats the last slab: harmless for a minimum)
      v[u] = *(const u4 *)(keys + (size_t)c * slab_stride + 4 * t);
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      uint32_t a = v[u].x < v[u].y ? v[u].x : v[u].y, b = v[u].z < v[u].w ? v[u].z : v[u].w;
      a = a < b ? a : b;
      m = m < a ? m : a;
    }
  }
  // K-th smallest of the 1024 group minima: four 8-bit radix levels over LDS histograms (a rank-by-counting
  // pass cost 1M comparisons on this one CU = 40 us; this is ~1 us)
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_prefix, sh_krem;
  if (t == 0) {
    sh_prefix = 0;
    sh_krem = k;
  }
  for (int p = 0; p < 4; p++) {
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const uint32_t prefix = sh_prefix, krem = sh_krem;
    const int shift = 24 - 8 * p;
    if (p == 0 || (m >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(m >> shift) & 255u], 1u);
    __syncthreads();
    if (t < 64) {  // wavefront 0: inclusive scan of the 256 bins, 4 per lane
      const uint32_t h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
      const uint32_t sum = h0 + h1 + h2 + h3;
      uint32_t inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (t >= (uint32_t)off) inc += o;
      }
      const uint32_t exc = inc - sum;
      if (exc < krem && krem <= inc) {  // exactly one lane owns the digit of the K-th element
        uint32_t c = exc, d = 0;
        if (krem > c + h0) { c += h0; d = 1; if (krem > c + h1) { c += h1; d = 2; if (krem > c + h2) { c += h2; d = 3; } } }
        sh_prefix = prefix | ((4 * t + d) << shift);
        sh_krem = krem - c;
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    const uint32_t kth = sh_prefix;
    const uint32_t u = (kth & 0x80000000u) ? (kth ^ 0x80000000u) : ~kth;
    tau_out[0] = __uint_as_float(u);
  }
}

// One streaming pass: (row, key) of every key <= orderable(*tau).  Four independent 16-byte loads per
// thread and step, ONE ballot per step on the minimum of the 16 keys (hits are rare), then the appends.
// (grid.y > 1: query b = blockIdx.y reads keys + b * keys_ld and tau[b], appends to cand + b * cap counting in cand_count[b])
__global__ __launch_bounds__(256) void filter_keys_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                          const float *__restrict__ tau, uint2 *__restrict__ cand,
                                                          uint32_t *__restrict__ cand_count, uint32_t cap, float slack,
                                                          uint32_t keys_ld, const float *__restrict__ slack_q) {
  keys += (size_t)blockIdx.y * keys_ld;
  cand += (size_t)blockIdx.y * cap;
  cand_count += blockIdx.y;
  const uint32_t max_key = f2key_dev(tau[blockIdx.y] + (slack_q ? slack_q[blockIdx.y] : slack));
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t n4 = n / 4;  // whole 16-byte chunks; the 0..3 keys behind them are handled at the end
  const u4 *k4 = (const u4 *)keys;
  auto append = [&](bool take, uint32_t row, uint32_t key) {
    u64 m = __ballot(take);
    if (!m) return;
    int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == (uint32_t)leader) base = atomicAdd(cand_count, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (take) {
      uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (slot < cap) cand[slot] = make_uint2(row, key);
    }
  };
  for (uint32_t b0 = blockIdx.x * 1024; b0 < n4; b0 += gridDim.x * 1024) {  // wave-uniform trip count
    u4 v[4];
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t idx = b0 + u * 256 + threadIdx.x;
      v[u] = idx < n4 ? __builtin_nontemporal_load(k4 + idx) : (u4){~0u, ~0u, ~0u, ~0u};
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      uint32_t a = v[u].x < v[u].y ? v[u].x : v[u].y, b = v[u].z < v[u].w ? v[u].z : v[u].w;
      a = a < b ? a : b;
      mn = mn < a ? mn : a;
    }
    if (__ballot(mn <= max_key)) {
      // one atomic per wavefront and step: lanes count their own hits, a shuffle scan hands out the slots
      uint32_t mine = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t idx = b0 + u * 256 + threadIdx.x;
        if (idx < n4)
          mine += (v[u].x <= max_key ? 1u : 0u) + (v[u].y <= max_key ? 1u : 0u) + (v[u].z <= max_key ? 1u : 0u) +
                  (v[u].w <= max_key ? 1u : 0u);
      }
      uint32_t incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += t;
      }
      const uint32_t total = __shfl(incl, 63, 64);
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(cand_count, total);
      base = __shfl(base, 0, 64);
      if (mine) {
        uint32_t slot = base + incl - mine;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t idx = b0 + u * 256 + threadIdx.x;
          const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (idx < n4 && w[j] <= max_key) {
              if (slot < cap) cand[slot] = make_uint2(idx * 4 + j, w[j]);
              slot++;
            }
        }
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const uint32_t row = n4 * 4 + threadIdx.x;
    const uint32_t key = row < n ? keys[row] : 0xFFFFFFFFu;
    append(row < n && key <= max_key, row, key);
  }
}

inline uint32_t pass_grid(uint32_t n) {
  uint32_t need = ((n + 3) / 4 + 255) / 256;
  uint32_t cap = (uint32_t)scan_tuning().num_cus * 4;
  uint32_t g = need < cap ? need : cap;
  return g ? g : 1;
}

}  // namespace

void launch_select_pass(const void *keys, int key_bytes, uint32_t n, int pass, uint32_t k, uint64_t lkey,
                        uint32_t lrow, int has_lower, const SelectBufs &b, hipStream_t s) {
  if (key_bytes == 8)
    hipLaunchKernelGGL(select_pass_kernel<u64>, dim3(pass_grid(n)), dim3(256), 0, s, (const u64 *)keys, n, pass, k,
                       (u64)lkey, lrow, has_lower, b.hist);
  else
    hipLaunchKernelGGL(select_pass_kernel<uint32_t>, dim3(pass_grid(n)), dim3(256), 0, s, (const uint32_t *)keys, n,
                       pass, k, (u64)lkey, lrow, has_lower, b.hist);
}

void launch_select_collect(const void *keys, int key_bytes, uint32_t n, int passes_done, uint32_t k, uint64_t lkey,
                           uint32_t lrow, int has_lower, const SelectBufs &b, uint32_t cap, hipStream_t s) {
  if (key_bytes == 8)
    hipLaunchKernelGGL(select_collect_kernel<u64>, dim3(pass_grid(n)), dim3(256), 0, s, (const u64 *)keys, n,
                       passes_done, k, (u64)lkey, lrow, has_lower, b.hist, b.counters, b.out_rows, (u64 *)b.out_keys,
                       (u64 *)b.bound, cap);
  else
    hipLaunchKernelGGL(select_collect_kernel<uint32_t>, dim3(pass_grid(n)), dim3(256), 0, s, (const uint32_t *)keys,
                       n, passes_done, k, (u64)lkey, lrow, has_lower, b.hist, b.counters, b.out_rows,
                       (uint32_t *)b.out_keys, (u64 *)b.bound, cap);
}

void launch_sample_threshold(const uint32_t *keys, uint32_t n, uint32_t per, uint32_t k, float *tau_out,
                             uint32_t *zero4, hipStream_t s) {
  const uint32_t slabs = per / 4;  // caller guarantees n >= 1024 * per (>= 4096 * slabs)
  const uint32_t slab_stride = slabs > 1 ? ((n - 4096) / (slabs - 1)) & ~3u : 0;
  hipLaunchKernelGGL(sample_threshold_kernel, dim3(1), dim3(1024), 0, s, keys, n, slab_stride, slabs, k, tau_out,
                     zero4, 0u, (uint32_t *)nullptr);
}

void launch_sample_threshold_batch(const uint32_t *keys, uint32_t keys_ld, uint32_t n, uint32_t per, uint32_t k,
                                   uint32_t n_queries, float *tau_out, uint32_t *cand_count, hipStream_t s) {
  const uint32_t slabs = per / 4;
  const uint32_t slab_stride = slabs > 1 ? ((n - 4096) / (slabs - 1)) & ~3u : 0;
  hipLaunchKernelGGL(sample_threshold_kernel, dim3(1, n_queries), dim3(1024), 0, s, keys, n, slab_stride, slabs, k, tau_out,
                     (uint32_t *)nullptr, keys_ld, cand_count);
}

void launch_filter_keys_batch(const uint32_t *keys, uint32_t keys_ld, uint32_t n, uint32_t n_queries, const float *tau,
                              void *cand, uint32_t *cand_count, uint32_t cap, hipStream_t s, const float *slack_q) {
  uint32_t need = (n / 4 + 1023) / 1024;
  hipLaunchKernelGGL(filter_keys_kernel, dim3(need ? need : 1, n_queries), dim3(256), 0, s, keys, n, tau, (uint2 *)cand,
                     cand_count, cap, 0.0f, keys_ld, slack_q);
}

void launch_filter_keys(const uint32_t *keys, uint32_t n, const float *tau, void *cand, uint32_t *cand_count,
                        uint32_t cap, hipStream_t s, float slack) {
  // one step per workgroup (a capped grid left ~20 % of the workgroups a second step: 21 us instead of ~12)
  uint32_t need = (n / 4 + 1023) / 1024;
  hipLaunchKernelGGL(filter_keys_kernel, dim3(need ? need : 1), dim3(256), 0, s, keys, n, tau, (uint2 *)cand, cand_count, cap,
                     slack, 0u, (const float *)nullptr);
}

namespace {
__global__ __launch_bounds__(256) void cand_rows_kernel(const uint2 *__restrict__ cand, const uint32_t *__restrict__ count,
                                                        uint32_t cap, uint32_t *__restrict__ rows_out) {
  const uint32_t m = count[0] < cap ? count[0] : cap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) rows_out[i] = cand[i].x;
}
__global__ __launch_bounds__(256) void cand_set_keys_kernel(uint2 *__restrict__ cand, const float *__restrict__ dists,
                                                            uint32_t m) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) cand[i].y = f2key_dev(dists[i]);
}
}  // namespace
void launch_cand_rows(const void *cand, const uint32_t *count, uint32_t cap, uint32_t *rows_out, hipStream_t s) {
  hipLaunchKernelGGL(cand_rows_kernel, dim3(64), dim3(256), 0, s, (const uint2 *)cand, count, cap, rows_out);
}
void launch_cand_set_keys(void *cand, const float *dists, uint32_t m, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(cand_set_keys_kernel, dim3((m + 255) / 256), dim3(256), 0, s, (uint2 *)cand, dists, m);
}

void launch_range(const void *keys, int key_bytes, uint32_t n, uint64_t max_key, int collect, uint32_t *counters,
                  uint32_t *out_rows, void *out_keys, uint32_t cap, hipStream_t s) {
  uint32_t need = (n + 255) / 256, cap_g = (uint32_t)scan_tuning().num_cus * 4;
  uint32_t g = need < cap_g ? need : cap_g;
  if (key_bytes == 8)
    hipLaunchKernelGGL(range_kernel<uint64_t>, dim3(g ? g : 1), dim3(256), 0, s, (const uint64_t *)keys, n, max_key,
                       collect, counters, out_rows, (uint64_t *)out_keys, cap);
  else
    hipLaunchKernelGGL(range_kernel<uint32_t>, dim3(g ? g : 1), dim3(256), 0, s, (const uint32_t *)keys, n,
                       (uint32_t)max_key, collect, counters, out_rows, (uint32_t *)out_keys, cap);
}

}  // namespace rsgpu
