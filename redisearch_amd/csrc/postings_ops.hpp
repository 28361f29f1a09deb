// postings_ops.hpp -- device pieces shared by postings_kernels.hip (the staged intersection / scoring / selection kernels)
// and hybrid_kernels.hip (the same work for a whole query in two launches): the searches of the intersection probe, the
// wave-wide top-k, and the scorers.  Both files are compiled with -ffp-contract=off and use THESE definitions, so a
// document's score has the same bits whichever kernel computed it.
#pragma once
#include <hip/hip_runtime.h>

#include "scan_ops.hpp"  // f2key / d2key, the chunk operators of the KNN branch
#include "search_kernels.hpp"

namespace rsgpu {
namespace {

// ---- intersection ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound(const uint32_t *__restrict__ a, uint32_t n, uint32_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < x) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// lower_bound by a whole wavefront: 64 evenly spaced probes per step narrow [lo, hi) 64-fold, so a 5 M-entry list takes 3 dependent
// memory round trips instead of 23 (the probes of one step are independent loads) -- WITHOUT a last probe that would pin the
// position down: *lo_out <= lower_bound(x) <= *hi_out, at most 64 apart, because the callers stage a window anyway and a window
// up to 64 entries wider at either end costs nothing, while the last probe is one more dependent round trip; and with the first
// level's probes -- 64 positions that depend on the list's length alone -- REQUESTED before the value searched for is known
// (use_first: `first` = a[(lane + 1) * ceil(n / 64) - 1], anything past the end), so that a tile's first doc id and the probes
// travel together.
__device__ __forceinline__ void wave_lower_bound_range(const uint32_t *__restrict__ a, uint32_t n, uint32_t x, uint32_t lane,
                                                       uint32_t first, bool use_first, uint32_t *lo_out, uint32_t *hi_out) {
  uint32_t lo = 0, hi = n;  // answer in [lo, hi]
  bool level1 = use_first;
  while (hi - lo > 64) {
    const uint32_t step = (hi - lo + 63) / 64;
    const uint32_t p = lo + (lane + 1) * step - 1;
    const uint32_t v = level1 ? first : a[p < hi ? p : hi - 1];
    level1 = false;
    const bool less = p < hi ? v < x : false;
    const uint32_t c = (uint32_t)__popcll(__ballot(less));
    const uint32_t nlo = lo + c * step;
    const uint32_t nhi = nlo + step - 1 < hi ? nlo + step - 1 : hi;
    lo = nlo < hi ? nlo : hi;
    hi = nhi;
  }
  *lo_out = lo;
  *hi_out = hi;
}

// id i of list l in the frame the lists of a query share
__device__ __forceinline__ uint32_t shared_id(const ListView &v, int l, uint32_t i) {
  return (uint32_t)((long long)v.ids[l][i] + v.add[l]);
}
// x (shared frame) -> the frame of a list stored `add` away from it; *out: x lies outside the 32-bit range that list
// can hold -- it cannot match, and the value returned keeps its lower bound right (0 below, the list's end above)
__device__ __forceinline__ uint32_t to_list_frame(uint32_t x, long long add, bool *out) {
  const long long t = (long long)x - add;
  *out = t < 0 || t > 0xFFFFFFFFll;
  return t < 0 ? 0u : (t > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)t);
}

// ---- wave-wide top-k ---------------------------------------------------------------------------------
constexpr int kKnnTopkBlocks = 64, kKnnTopkPerThread = 4, kKnnTopkMaxK = 32;
// wave-wide minimum, the same value in every lane.  Four DPP steps (xor 1, xor 2 inside a quad, half-row mirror, row
// mirror: min is idempotent, so mirrors do as well as butterflies) leave every lane with the minimum of its row of 16;
// four v_readlane pairs and scalar compares finish it -- ~40 instructions instead of twelve dependent ds_bpermute round
// trips (the k rounds of wave_topk are a serial chain of these).
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_min_step(uint64_t v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
  const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
  const uint64_t o = ((uint64_t)ohi << 32) | olo;
  return o < v ? o : v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
  v = dpp_min_step<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_min_step<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_min_step<0x141>(v);  // row_half_mirror
  v = dpp_min_step<0x140>(v);  // row_mirror
  uint64_t m = ~0ull;
#pragma unroll
  for (int row = 0; row < 4; row++) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, row * 16);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), row * 16);
    const uint64_t r = ((uint64_t)hi << 32) | lo;
    m = r < m ? r : m;
  }
  return m;
}
// the k (<= 64) smallest of the N composites each lane holds: lane r returns the r-th smallest (~0 when there are
// fewer).  k rounds of a wave-wide arg-min -- shuffles only, no barrier, no LDS.  Composites are unique.
template <int N>
__device__ __forceinline__ uint64_t wave_topk(uint64_t (&mine)[N], uint32_t k, uint32_t lane) {
  uint64_t res = ~0ull;
  for (uint32_t r = 0; r < k; r++) {
    uint64_t v = mine[0];
#pragma unroll
    for (int j = 1; j < N; j++) v = mine[j] < v ? mine[j] : v;
    const uint64_t m = wave_min_u64(v);
    if (m == ~0ull) break;
    if (lane == r) res = m;
#pragma unroll
    for (int j = 0; j < N; j++)
      if (mine[j] == m) mine[j] = ~0ull;
  }
  return res;
}

// ---- proximity: max_slop / in_order and the scorers' slop, over the offset bytes in place ---------------------------------
// Reference: index_result/src/core/proximity.rs:134-298 (within_range_in_order / _unordered, OffsetIter::Merge) and
// src/index_result/index_result.c:51-103 (IndexResult_MinOffsetDelta).  One thread per candidate / hit; a term's
// positions are read straight out of the list's byte buffer (varint deltas), nothing is materialised.
constexpr uint32_t kPosEof = 0xFFFFFFFFu;
struct TermIt {
  const uint8_t *p;
  uint32_t len, pos, last;
};
__device__ __forceinline__ uint32_t term_next(TermIt &t) {
  if (t.pos >= t.len) return kPosEof;
  uint32_t c = t.p[t.pos++];
  uint32_t val = c & 0x7fu;
  while (c & 0x80u) {
    if (t.pos >= t.len) return kPosEof;  // truncated varint: the reference's reader errors -> EOF
    val++;
    c = t.p[t.pos++];
    val = (val << 7) | (c & 0x7fu);
  }
  t.last += val;
  return t.last;
}

template <int MAXL>
struct ProxCtx {
  TermIt leaf[MAXL];
  uint32_t look[MAXL];
  uint32_t present;  // bit l: leaf l matched this document (union children may be absent)

  __device__ __forceinline__ bool child_present(const ProxParams &P, int c) const {
    uint32_t m = 0;
    for (int l = P.child_first[c]; l < P.child_first[c + 1]; l++) m |= (present >> l) & 1u;
    return m != 0;
  }

  __device__ __forceinline__ bool merged(const ProxParams &P, int c) const {
    return P.is_agg[c] && (P.child_first[c + 1] - P.child_first[c]) != 1;
  }
  // proximity.rs:72-90: a term takes part iff it has offsets, an aggregate of terms always does
  __device__ __forceinline__ bool has(const ProxParams &P, int c) const {
    return P.is_agg[c] ? child_present(P, c) : (P.child_first[c + 1] > P.child_first[c] && leaf[P.child_first[c]].len > 0);
  }
  __device__ __forceinline__ void reset(const ProxParams &P, int c) {
    for (int l = P.child_first[c]; l < P.child_first[c + 1]; l++) {
      leaf[l].pos = 0;
      leaf[l].last = 0;
    }
    if (merged(P, c))
      for (int l = P.child_first[c]; l < P.child_first[c + 1]; l++) look[l] = term_next(leaf[l]);
  }
  __device__ __forceinline__ uint32_t next(const ProxParams &P, int c) {
    const int a = P.child_first[c], b = P.child_first[c + 1];
    if (!merged(P, c)) return b > a ? term_next(leaf[a]) : kPosEof;
    int best = -1;
    uint32_t mv = kPosEof;
    for (int l = a; l < b; l++)
      if (look[l] != kPosEof && look[l] < mv) {
        mv = look[l];
        best = l;
      }
    if (best < 0) return kPosEof;
    look[best] = term_next(leaf[best]);
    return mv;
  }
};

template <int MAXL>
__device__ bool prox_within_range(const ProxParams &P, ProxCtx<MAXL> &x) {
  if (P.n_children <= 1) return true;
  int m[MAXL], n = 0;
  for (int c = 0; c < P.n_children; c++)
    if (x.has(P, c)) {
      x.reset(P, c);
      m[n++] = c;
    }
  if (n <= 1) return true;
  const uint32_t max_slop = P.max_slop < 0 ? 0xFFFFFFFFu : (uint32_t)P.max_slop;
  uint32_t positions[MAXL];
  if (P.in_order) {
    for (int i = 0; i < n; i++) positions[i] = 0;
    for (;;) {
      int span = 0;
      bool over = false;
      for (int i = 0; i < n; i++) {
        uint32_t pos;
        if (i == 0) {
          pos = x.next(P, m[0]);
          if (pos == kPosEof) return false;
        } else {
          pos = positions[i];
        }
        const uint32_t last_pos = i == 0 ? 0u : positions[i - 1];
        while (pos < last_pos) {
          pos = x.next(P, m[i]);
          if (pos == kPosEof) return false;
        }
        positions[i] = pos;
        if (i > 0) {
          span += (int)pos - (int)last_pos - 1;
          if (span > 0 && (uint32_t)span > max_slop) {
            over = true;
            break;
          }
        }
      }
      if (!over) return true;
    }
  }
  for (int i = 0; i < n; i++) {
    positions[i] = x.next(P, m[i]);
    if (positions[i] == kPosEof) return false;
  }
  uint32_t max_pos = 0;
  for (int i = 0; i < n; i++)
    if (positions[i] >= max_pos) max_pos = positions[i];
  for (;;) {
    uint32_t min_pos = kPosEof;
    int min_idx = 0;
    for (int i = 0; i < n; i++)
      if (positions[i] < min_pos) {
        min_pos = positions[i];
        min_idx = i;
      }
    if (min_pos != max_pos) {
      const int span = (int)max_pos - (int)min_pos - (n - 1);
      if (span < 0 || (uint32_t)span <= max_slop) return true;
    }
    const uint32_t np = x.next(P, m[min_idx]);
    if (np == kPosEof) return false;
    positions[min_idx] = np;
    if (np > max_pos) max_pos = np;
  }
}

template <int MAXL>
__device__ int prox_min_offset_delta(const ProxParams &P, ProxCtx<MAXL> &x) {
  const int nc = P.n_children;
  // a union's aggregate only holds the children that matched this document (union_flat.rs:297-320)
  int num = nc;
  if (P.count_present) {
    num = 0;
    for (int c = 0; c < nc; c++) num += x.child_present(P, c) ? 1 : 0;
  }
  if (num <= 1) return 1;
  int dist = 0, i = 0;
  while (i < nc) {
    while (i < nc && !x.has(P, i)) i++;
    if (i == nc) break;
    const int c1 = i++;
    while (i < nc && !x.has(P, i)) i++;
    if (i == nc) break;
    const int c2 = i;  // (not consumed: it is the first of the next pair)
    x.reset(P, c1);
    x.reset(P, c2);
    uint32_t p1 = x.next(P, c1), p2 = x.next(P, c2);
    int cd = (int)(p2 > p1 ? p2 - p1 : p1 - p2);
    while (cd > 1 && p1 != kPosEof && p2 != kPosEof) {
      const uint32_t a = p2 > p1 ? p2 - p1 : p1 - p2;
      if (a < (uint32_t)cd) cd = (int)a;
      if (p2 > p1) p1 = x.next(P, c1);
      else p2 = x.next(P, c2);
    }
    dist += cd * cd;
  }
  return dist ? (int)sqrt((double)dist) : num - 1;
}

// ---- two plain terms (round 5): `"hello world"`, `hello world` with SLOP -- the commonest windowed query ----------------------
// prox_within_range / prox_min_offset_delta restated for exactly two children that are single terms, with the two cursors and the
// two positions in REGISTERS: the general forms index their cursor arrays with run-time child / leaf numbers, which puts them in
// scratch memory (240 bytes per lane in hybrid_tree_tile_kernel, a memory round trip per position step).  Same walks, same
// comparisons, same results (proximity.rs:134-298, index_result.c:51-103); a term takes part iff it has offsets (len > 0).
__device__ __forceinline__ bool prox_two_terms(const ProxParams &P) {
  return P.n_children == 2 && P.n_leaves == 2 && !P.is_agg[0] && !P.is_agg[1] && P.child_first[0] == 0 && P.child_first[1] == 1 &&
         P.child_first[2] == 2;
}
__device__ __forceinline__ bool prox_within_range2(const ProxParams &P, TermIt a, TermIt b) {
  if (a.len == 0 || b.len == 0) return true;  // fewer than two children carry positions
  const uint32_t max_slop = P.max_slop < 0 ? 0xFFFFFFFFu : (uint32_t)P.max_slop;
  if (P.in_order) {
    uint32_t p1 = 0;
    for (;;) {
      const uint32_t p0 = term_next(a);
      if (p0 == kPosEof) return false;
      uint32_t pos = p1;
      while (pos < p0) {
        pos = term_next(b);
        if (pos == kPosEof) return false;
      }
      p1 = pos;
      const int span = (int)p1 - (int)p0 - 1;
      if (span > 0 && (uint32_t)span > max_slop) continue;
      return true;
    }
  }
  uint32_t p0 = term_next(a);
  if (p0 == kPosEof) return false;
  uint32_t p1 = term_next(b);
  if (p1 == kPosEof) return false;
  uint32_t max_pos = p1 >= p0 ? p1 : p0;
  for (;;) {
    const bool first = !(p1 < p0);  // (the smaller position; child 0 on a tie)
    const uint32_t min_pos = first ? p0 : p1;
    if (min_pos != max_pos) {
      const int span = (int)max_pos - (int)min_pos - 1;
      if (span < 0 || (uint32_t)span <= max_slop) return true;
    }
    const uint32_t np = first ? term_next(a) : term_next(b);
    if (np == kPosEof) return false;
    if (first) p0 = np;
    else p1 = np;
    if (np > max_pos) max_pos = np;
  }
}
__device__ __forceinline__ int prox_min_offset_delta2(TermIt a, TermIt b) {
  if (a.len == 0 || b.len == 0) return 1;  // (no pair of children with positions: children - 1)
  uint32_t p1 = term_next(a), p2 = term_next(b);
  int cd = (int)(p2 > p1 ? p2 - p1 : p1 - p2);
  while (cd > 1 && p1 != kPosEof && p2 != kPosEof) {
    const uint32_t d = p2 > p1 ? p2 - p1 : p1 - p2;
    if (d < (uint32_t)cd) cd = (int)d;
    if (p2 > p1) p1 = term_next(a);
    else p2 = term_next(b);
  }
  const int dist = cd * cd;
  return dist ? (int)sqrt((double)dist) : 1;
}
template <typename View>
__device__ __forceinline__ TermIt prox_term(const View &o, int leaf, uint32_t entry) {
  const bool on = o.off_pos[leaf] != nullptr && entry != 0xFFFFFFFFu;
  return TermIt{on ? o.bytes[leaf] + o.off_pos[leaf][entry] : nullptr, on ? o.off_len[leaf][entry] : 0u, 0u, 0u};
}

// (View: OffsetView, or any struct with the same three arrays over fewer leaves)
template <int MAXL, typename View, typename EntryOf>
__device__ __forceinline__ void prox_load(const ProxParams &P, const View &o, ProxCtx<MAXL> &x, EntryOf entry_of) {
  x.present = 0;
  for (int l = 0; l < P.n_leaves; l++) {
    const uint32_t e = entry_of(l);
    if (e != 0xFFFFFFFFu) x.present |= 1u << l;
    const bool on = o.off_pos[l] != nullptr && e != 0xFFFFFFFFu;
    x.leaf[l].p = on ? o.bytes[l] + o.off_pos[l][e] : nullptr;
    x.leaf[l].len = on ? o.off_len[l][e] : 0u;
    x.leaf[l].pos = 0;
    x.leaf[l].last = 0;
  }
}

// ---- scorers ---------------------------------------------------------------------------------------
// One document's score.  F(t): frequency of term column t in this document (fp64); dlen / dscore / mfreq: its doc-table
// entry (0 when the table does not hold it); slop: IndexResult_MinOffsetDelta of the result.
// DEEP: the tree is deeper than root -> groups -> leaves (P.n_nodes > 0); its per-level accumulators are indexed
// dynamically and live in scratch memory: the flat / two-level form must not pay for them.
// The result tree (ScoreParams): root -> groups -> leaves.  fold(leaf) evaluates it the way the reference's recursions do
// (src/ext/default.c:68-106,164-209,262-302,378-455): an aggregate sums its children and multiplies by its weight; DISMAX
// takes the maximum over a UNION's children instead.  A leaf that did not match this document (union children) carries
// frequency 0 and contributes exactly 0.
// FLAT (> 0): the caller knows the tree is an intersection of FLAT-or-fewer TERMS (every group is one leaf, no union
// anywhere): the same sum in the same order -- 0.0 + leaf(0) + leaf(1) ... -- with the leaf numbers compile-time constants,
// so the weights / idfs come out of the argument block with constant offsets, all at once, instead of one dependent scalar
// load after the other.
// MAXD (DEEP only): the deepest node the caller admits.  Up to 8 levels the accumulators are picked by compare-and-select over
// constant indices and stay in registers (the hybrid tile kernel's form: no scratch); beyond, they are indexed dynamically.
// SP: ScoreParams, or -- FLAT > 0 only -- ScoreParamsFlat (search_kernels.hpp: the fields a flat AND of <= FLAT terms reads).
template <bool DEEP, int FLAT = 0, int MAXD = kMaxTreeDepth, typename SP, typename FreqFn>
__device__ __forceinline__ double score_one(const SP &P, FreqFn F, uint32_t dlen, float dscore, uint32_t mfreq,
                                            int slop) {
  double s = 0.0;
  auto fold = [&](auto leaf, bool dismax) {
    if constexpr (FLAT > 0) {
      double ret = 0.0;
#pragma unroll
      for (int g = 0; g < FLAT; g++)
        if (g < P.n_groups) ret = ret + leaf(g);
      return ret;
    } else if constexpr (DEEP) {
      // any depth: one accumulator per open level.  Post-order: when an aggregate comes up, acc[its depth] holds the
      // sum (DISMAX under a union: the maximum) of its children, in the result's child order -- the order the
      // reference's recursions add them in -- and its own value, weight * that, goes to its parent's accumulator.
      double acc[MAXD + 1];
#pragma unroll
      for (int d = 0; d <= MAXD; d++) acc[d] = 0.0;
      for (int i = 0; i < P.n_nodes - 1; i++) {
        const int d = P.node_depth[i];
        double v;
        if constexpr (MAXD <= 8) {
          double own = 0.0, up = 0.0;
#pragma unroll
          for (int q = 1; q <= MAXD; q++) {
            own = q == d ? acc[q] : own;
            up = q == d ? acc[q - 1] : up;
          }
          const bool agg = P.node_op[i] != 0;
          v = agg ? P.node_weight[i] * own : leaf((int)P.node_leaf[i]);
          const double nu = (dismax && P.node_in_union[i]) ? (v > up ? v : up) : up + v;
#pragma unroll
          for (int q = 1; q <= MAXD; q++) {
            if (q == d && agg) acc[q] = 0.0;
            if (q == d) acc[q - 1] = nu;
          }
        } else {
          if (P.node_op[i] == 0) {
            v = leaf((int)P.node_leaf[i]);
          } else {
            v = P.node_weight[i] * acc[d];
            acc[d] = 0.0;
          }
          acc[d - 1] = (dismax && P.node_in_union[i]) ? (v > acc[d - 1] ? v : acc[d - 1]) : acc[d - 1] + v;
        }
      }
      return acc[0];
    } else {
      double ret = 0.0;
      for (int g = 0; g < P.n_groups; g++) {
        const int a = P.group_first[g], b = P.group_first[g + 1];
        double child;
        if (P.group_op[g] == 0) {
          child = leaf(a);
        } else {
          double acc = 0.0;
          for (int t = a; t < b; t++) {
            const double v = leaf(t);
            acc = (dismax && P.group_op[g] == 1) ? (v > acc ? v : acc) : acc + v;
          }
          child = P.group_weight[g] * acc;
        }
        ret = (dismax && P.is_union) ? (child > ret ? child : ret) : ret + child;
      }
      return ret;
    }
  };
  switch (P.scorer) {
    case 0:    // BM25STD      reference src/ext/default.c:241-316
    case 1: {  // BM25STD.TANH reference src/ext/default.c:329-359
      const float b = 0.75f, k1 = 1.2f;
      double ret = fold([&](int t) {
        const double f = F(t);
        // weight * idf * f * (k1 + 1) / (f + k1 * (1.0f - b + b * (float)doc_len/avg_doc_len))
        const double num = P.weight[t] * P.bm25_idf[t] * f * (double)(k1 + 1);
        const double den = f + (double)k1 * ((double)(1.0f - b) + (double)(b * (float)(int)dlen) / P.avg_doc_len);
        return num / den;
      }, false);
      ret *= P.root_weight;
      s = (double)dscore * ret;
      if (P.scorer == 1) s = tanh(P.inv_tanh * s);
      break;
    }
    case 2: {  // legacy BM25 reference src/ext/default.c:164-233
      const float b = 0.5f, k1 = 1.2f;
      double ret = fold([&](int t) {
        const double f = F(t);
        return P.weight[t] * P.idf[t] * f / (f + (double)k1 * ((double)(1.0f - b) + (double)b * P.avg_doc_len));
      }, false);
      ret *= P.root_weight;
      s = (double)dscore * ret;
      if (s < P.min_score) s = 0.0;
      else s /= (double)slop;
      break;
    }
    case 3:    // TFIDF         reference src/ext/default.c:109-145
    case 4: {  // TFIDF.DOCNORM reference src/ext/default.c:149-153
      const uint32_t norm = P.scorer == 3 ? mfreq : dlen;
      if (dscore == 0.0f || norm == 0) { s = 0.0; break; }
      double raw = fold([&](int t) { return P.weight[t] * F(t) * P.idf[t]; }, false);
      raw *= P.root_weight;
      s = (double)dscore * raw / (double)norm;
      if (s < P.min_score) s = 0.0;
      else s /= (double)slop;
      break;
    }
    case 5:  // DOCSCORE reference src/ext/default.c:366-371
      s = (double)dscore;
      break;
    default: {  // DISMAX reference src/ext/default.c:378-461: an intersection sums its children, a union takes their maximum
      s = P.root_weight * fold([&](int t) { return P.weight[t] * F(t); }, true);
      break;
    }
  }
  return s;
}

}  // namespace
}  // namespace rsgpu
