// search_abi.cpp -- host drivers + C ABI (include/rsgpu_search.h) of the integer / scoring half:
// device-resident posting lists in the reference's block format, GPU decode + N-way intersection,
// the built-in scorers over the hits, score top-N and the hybrid ad-hoc KNN step.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>
#include <thread>

#include "flat_index.hpp"
#include "rsgpu_search.h"
#include "search_kernels.hpp"

using namespace rsgpu;

namespace {
// one step of a polling loop (the completion flags of the hybrid kernels in pinned memory): the x86 pause hint where there is
// one, a yield elsewhere -- the library builds on any host (round-4 advisor)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#else
  std::this_thread::yield();
#endif
}

thread_local double prof_ms[5] = {0, 0, 0, 0, 0};  // decode, intersect, score, topn, knn

// Device buffers of the hit lists come and go with every query: hipMalloc costs tens of microseconds and
// hipFree synchronises the whole device, so freed buffers are parked in a per-device pool by power-of-two size
// class (at most kPoolCap bytes parked; beyond that they really are freed).
class DevPool {
 public:
  static DevPool &get() {
    static DevPool *p = new DevPool();  // leaked on purpose: outlives static destructors
    return *p;
  }
  static size_t size_class(size_t bytes) {
    size_t c = 4096;
    while (c < bytes) c <<= 1;
    return c;
  }
  void *take(size_t bytes, size_t *got, int *dev_out) {
    const size_t c = size_class(bytes);
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    *dev_out = dev;
    {
      std::lock_guard<std::mutex> g(mu_);
      auto &v = free_[key(dev, c)];
      if (!v.empty()) {
        void *p = v.back();
        v.pop_back();
        parked_ -= c;
        *got = c;
        return p;
      }
    }
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, c));
    *got = c;
    return p;
  }
  void drain() {
    std::unordered_map<uint64_t, std::vector<void *>> f;
    {
      std::lock_guard<std::mutex> g(mu_);
      f.swap(free_);
      parked_ = 0;
    }
    for (auto &kv : f)
      for (void *p : kv.second) (void)hipFree(p);
  }
  void give(void *p, size_t cls, int dev) {  // dev: the device the buffer was allocated on
    {
      std::lock_guard<std::mutex> g(mu_);
      if (parked_ + cls <= kPoolCap) {
        free_[key(dev, cls)].push_back(p);
        parked_ += cls;
        return;
      }
    }
    (void)hipFree(p);
  }

 private:
  static constexpr size_t kPoolCap = 4ull << 30;
  static uint64_t key(int dev, size_t c) { return ((uint64_t)dev << 56) | (uint64_t)c; }
  std::mutex mu_;
  std::unordered_map<uint64_t, std::vector<void *>> free_;
  size_t parked_ = 0;
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  size_t cls = 0;  // pool size class in bytes
  int dev = 0;     // device it lives on
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { reset(); }
  void reset() {
    if (p) DevPool::get().give(p, cls, dev);
    p = nullptr;
    n = 0;
    cls = 0;
  }
  void alloc(size_t count) {
    reset();
    if (!count) count = 1;
    p = static_cast<T *>(DevPool::get().take(count * sizeof(T), &cls, &dev));
    n = count;
  }
  void ensure(size_t count) {
    if (count > n) alloc(count + count / 4 + 64);
  }
  void upload(const T *src, size_t count) {
    alloc(count);
    if (count) HIP_CHECK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  }
};

// grow-only per-thread scratch for the intersection
struct Scratch {
  int device = -1;
  DevBuf<uint8_t> flags;
  DevBuf<uint32_t> pos, block_counts, total, rows, keys32, skeys32;
  DevBuf<float> dists;
  DevBuf<uint64_t> fuse, maxkey;
  // fused hybrid query, KNN branch: {candidate count, ticket} -- zero at rest, the last kernel of the branch puts them
  // back -- and the per-block partial lists of knn_topk_kernel
  DevBuf<uint32_t> knn_cnt;
  DevBuf<uint64_t> knn_part;
  bool knn_dirty = true;  // the counters may be non-zero (first use, or a query that failed half-way)
  // hybrid query in two launches (hybrid_kernels.hip): the tiles' lists, the reduce blocks' lists, the two tickets
  DevBuf<uint32_t> hyb_hits, hyb_sidx;
  DevBuf<uint64_t> hyb_skey, hyb_knn, hyb_trace, hyb_tie;
  uint32_t hyb_trace_tiles = 0;
  // ... its general form, hit list wanted: doc id | frequencies | entry indices at the tiles' fixed slots (hybrid_hits_pack)
  DevBuf<uint32_t> hyb_hit_ids, hyb_hit_freqs, hyb_hit_epos;
  DevBuf<uint32_t> hyb_run_ids, hyb_run_freqs, hyb_run_epos, hyb_run_start;  // the packed runs of a multi-pass query's hit list, before the merge
};
thread_local Scratch tls_scratch;
Scratch &scratch(int device) {
  Scratch &s = tls_scratch;
  if (s.device != device) {
    s.flags.reset(); s.pos.reset(); s.block_counts.reset(); s.total.reset(); s.rows.reset(); s.keys32.reset();
    s.skeys32.reset();
    s.dists.reset();
    s.fuse.reset();
    s.maxkey.reset();
    s.knn_cnt.reset();
    s.knn_part.reset();
    s.knn_dirty = true;
    s.hyb_hits.reset(); s.hyb_sidx.reset();
    s.hyb_skey.reset(); s.hyb_knn.reset(); s.hyb_trace.reset(); s.hyb_tie.reset();
    s.hyb_trace_tiles = 0;
    s.hyb_hit_ids.reset(); s.hyb_hit_freqs.reset(); s.hyb_hit_epos.reset();
    s.hyb_run_ids.reset(); s.hyb_run_freqs.reset(); s.hyb_run_epos.reset(); s.hyb_run_start.reset();
    s.device = device;
  }
  return s;
}

// per-stage device time; only when profiling is on (RSGPU_SetProfiling): each stop() synchronises the stream
struct StageTimer {
  QueryCtx *c;
  int slot;
  bool on;
  StageTimer(QueryCtx *ctx, int s) : c(ctx), slot(s), on(scan_profile().enabled.load(std::memory_order_relaxed) != 0) {
    if (on) HIP_CHECK(hipEventRecord(c->ev0, c->stream));
  }
  void stop() {
    if (!on) {  // the stage boundary stays a synchronisation point: results are read on other streams / the host
      HIP_CHECK(hipStreamSynchronize(c->stream));
      return;
    }
    HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    HIP_CHECK(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    prof_ms[slot] = ms;
  }
};

inline double key2score(uint64_t inv_key) {  // inverse of ~d2key(score)
  uint64_t k = ~inv_key;
  uint64_t u = (k & 0x8000000000000000ull) ? (k ^ 0x8000000000000000ull) : ~k;
  double d;
  memcpy(&d, &u, 8);
  return d;
}

}  // namespace

struct RSGPU_Postings {
  int device = 0, codec = 0;
  CodecDesc cd{};
  uint32_t n_blocks = 0, n_entries = 0;
  size_t n_bytes = 0;
  // 64-bit doc ids: the device arrays hold id - base (u32); base = 0 for lists whose ids all fit 32 bits, else the
  // list's first doc id.  first / last = the list's smallest / largest doc id (absolute).
  uint64_t base = 0, first_id = 0, last = 0;
  DevBuf<uint8_t> bytes;
  DevBuf<uint64_t> byte_off;
  DevBuf<uint32_t> first, nent, entry_off;
  DevBuf<uint32_t> ids, freqs, masks;  // decode targets
  DevBuf<uint32_t> wmasks;             // wide codecs: 128-bit field masks, 4 words per entry
  DevBuf<uint32_t> off_pos, off_len;   // codecs with offsets: where each entry's offsets blob sits in `bytes`
  bool has_offsets() const { return cd.osz >= 0; }
  // A list is immutable after upload, so its decoded arrays stay valid: with cache_decoded (default) the
  // decode kernel runs once, at the first query that touches the list, and HBM keeps both forms
  // (8-12 B per posting decoded next to ~3 B encoded).
  std::mutex decode_mu;
  std::atomic<bool> decoded{false};
  // ... of which the doc ids and frequencies alone (round 5): what the two-launch hybrid query reads.  A Full-codec list decodes
  // into 20 bytes per posting -- ids, frequencies, field masks, offsets position / length -- and a query that neither walks the
  // term offsets nor hands out term records needs 8 of them; the rest is decoded when a path first asks for it
  std::atomic<bool> decoded_lean{false};
  // qint layouts: sub-block sync points, left behind by the first decode for all later ones (search_kernels.hpp)
  DevBuf<uint32_t> sync;
  std::atomic<bool> sync_ready{false};
  uint32_t sync_span = 0;  // widest byte range of decode_sync_blocks_per_wave() consecutive blocks
  // bucket directory over the decoded doc ids (round 4; hybrid_kernels.hip): dir[b] = lower_bound(ids, b << dir_shift) --
  // built once by the first two-launch hybrid query that probes this list (a list is immutable: every later decode
  // reproduces the same ids), ~4 bytes per 32 postings
  DevBuf<uint32_t> dir;
  uint32_t dir_shift = 0, dir_n = 0;
  std::atomic<bool> dir_ready{false};
};

// one node of a hit list's result tree, post-order over its leaf columns (a term: op 0, `leaf`; an aggregate: op 1 union /
// 2 intersection over the n_children complete subtrees right before it)
struct TNode {
  uint8_t op = 0, leaf = 0;
  uint16_t n_children = 0;
  double weight = 1.0;
};

struct RSGPU_Hits {
  int device = 0, n_lists = 0;
  uint32_t len = 0, cap = 0;
  uint64_t base = 0, last = 0;  // ids[] hold doc id - base (base <= every id); last: an upper bound of the largest doc id
  int order[kMaxLists];  // internal list slot -> index in the caller's list array
  DevBuf<uint32_t> ids, freqs;
  DevBuf<double> scores;
  DevBuf<uint64_t> keys;
  bool scored = false;
  bool is_union = false;        // built by RSGPU_Union: absent lists have freq 0
  // term offsets: entry index of every hit in every list (slot order) + the lists themselves, so that the scorers'
  // slop (IndexResult_MinOffsetDelta) can be computed from the offset bytes; the lists must outlive the hits
  bool with_offsets = false;
  DevBuf<uint32_t> epos;
  DevBuf<int32_t> slops;
  bool slops_ready = false;
  const RSGPU_Postings *src[kMaxLists] = {};
  // result-tree shape for the scorers: the root's children ("groups") over the leaf columns; a flat list has one
  // term group per leaf
  int n_groups = 0;
  uint8_t group_first[kMaxLists + 1] = {};
  uint8_t group_op[kMaxLists] = {};  // 0 term, 1 union, 2 intersection
  double group_weight[kMaxLists] = {};
  std::vector<std::unique_ptr<RSGPU_Hits>> nested;  // the groups' own hit lists (their columns are borrowed)
  std::vector<TNode> tree;  // the whole result tree (any depth), post-order, root last; the groups above = the root's children
  std::vector<uint32_t> h_ids;  // lazily mirrored
  const std::vector<uint32_t> &host_ids() {
    if (h_ids.size() != len) {
      h_ids.resize(len);
      if (len) HIP_CHECK(hipMemcpy(h_ids.data(), ids.p, len * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return h_ids;
  }
};

struct RSGPU_DocTable {
  int device = 0;
  uint32_t n = 0;
  uint64_t first = 0;  // doc id of entry 0
  DevBuf<uint32_t> doc_len, max_freq;
  DevBuf<float> doc_score;
  DevBuf<uint64_t> len_score;  // {doc_len, bits of doc_score} per document: the two-launch hybrid query's one gather per hit
};

#define S_TRY try {
#define S_CATCH(failval)                         \
  }                                              \
  catch (const std::exception &e) {              \
    last_error() = e.what();                 \
    logf(nullptr, "warning", "%s", e.what());    \
    return failval;                              \
  }

// lean: doc ids + frequencies only (the field masks and the offsets index stay as they are)
static void launch_decode(RSGPU_Postings *p, QueryCtx *c, int sync_mode, bool lean = false) {
  const bool wide = p->cd.wide;  // (the wide codecs' masks travel with the generic loop: always whole)
  launch_decode_blocks(p->cd, p->bytes.p, p->byte_off.p, p->first.p, p->nent.p, p->entry_off.p, p->n_blocks, p->ids.p,
                       p->cd.freq >= 0 ? p->freqs.p : nullptr, ((p->cd.mask >= 0 || wide) && (!lean || wide)) ? p->masks.p : nullptr,
                       c->stream, wide ? p->wmasks.p : nullptr, (p->has_offsets() && (!lean || wide)) ? p->off_pos.p : nullptr,
                       (p->has_offsets() && (!lean || wide)) ? p->off_len.p : nullptr, p->sync.p, sync_mode, p->sync_span,
                       (uint32_t)std::min<uint64_t>(p->n_blocks ? p->n_bytes / p->n_blocks : 0, 0xFFFFFFFFull));
  HIP_CHECK(hipGetLastError());
}

// lean (round 5): the caller reads the doc ids and the frequencies only (hybrid_two_launches) -- see RSGPU_Postings::decoded_lean
static void decode_on(RSGPU_Postings *p, QueryCtx *c, bool force = false, bool lean = false) {
  const bool cached = scan_tuning().cache_decoded && !force;
  lean = lean && scan_tuning().decode_lean && !p->cd.wide;
  if (cached && (p->decoded.load(std::memory_order_acquire) || (lean && p->decoded_lean.load(std::memory_order_acquire)))) return;
  const bool has_sync = p->sync.p != nullptr && scan_tuning().decode_sync;
  if (has_sync && p->sync_ready.load(std::memory_order_acquire) && !cached) {  // eight lanes per block
    launch_decode(p, c, 2, lean);
    return;
  }
  if (!cached && !has_sync) {  // nothing to publish: no wait
    launch_decode(p, c, 0, lean);
    return;
  }
  // the decode that publishes something other streams will read -- the decoded arrays (cache) or the sync points
  // (UPGRADE of a lean decode -- decoded_lean set, a consumer of masks / offsets arrives: the full decode writes ids / freqs
  // AGAIN while other threads' tile kernels may be reading them, without a lock.  Sound by the identical-rewrite invariant: the
  // encoded bytes are immutable after RSGPU_Postings_Upload and the decode is a pure function of them, so every 4-byte store
  // stores the value that is already there; aligned 4-byte stores are atomic.  Nothing a reader can observe changes.)
  std::lock_guard<std::mutex> g(p->decode_mu);
  if (cached && (p->decoded.load(std::memory_order_relaxed) || (lean && p->decoded_lean.load(std::memory_order_relaxed)))) return;
  launch_decode(p, c, has_sync && !p->sync_ready.load(std::memory_order_relaxed) ? 1 : (has_sync ? 2 : 0), lean);
  HIP_CHECK(hipStreamSynchronize(c->stream));
  if (has_sync) p->sync_ready.store(true, std::memory_order_release);
  if (cached) {
    p->decoded_lean.store(true, std::memory_order_release);
    if (!lean) p->decoded.store(true, std::memory_order_release);
  }
}

// Decode-per-query mode (cache_decoded = 0): two qint lists whose sync points are there go up in ONE launch (the fixed cost
// of a launch is a third of a list's decode: profiles/r03_decode.txt).  false: not such a pair -- decode_on takes each.
static bool decode_pair_on(RSGPU_Postings *p, RSGPU_Postings *q, QueryCtx *c, bool lean = false) {
  if (scan_tuning().cache_decoded || !scan_tuning().decode_sync || !scan_tuning().decode_pair || p == q) return false;
  lean = lean && scan_tuning().decode_lean;
  for (RSGPU_Postings *x : {p, q})
    if (!x->sync.p || !x->sync_ready.load(std::memory_order_acquire) || !x->n_blocks || x->cd.kind != 0 || x->cd.wide) return false;
  auto args = [lean](RSGPU_Postings *x) {
    return DecodeListArgs{x->cd, x->bytes.p, x->byte_off.p, x->first.p, x->nent.p, x->entry_off.p, (uint32_t)x->n_blocks, x->ids.p,
                          x->cd.freq >= 0 ? x->freqs.p : nullptr, (x->cd.mask >= 0 && !lean) ? x->masks.p : nullptr, nullptr,
                          (x->has_offsets() && !lean) ? x->off_pos.p : nullptr, (x->has_offsets() && !lean) ? x->off_len.p : nullptr, x->sync.p, 2,
                          x->sync_span};
  };
  const bool ok = launch_decode_blocks_pair(args(p), args(q), c->stream);
  HIP_CHECK(hipGetLastError());
  return ok;
}

namespace rsgpu {
void release_search_pool() { DevPool::get().drain(); }
}  // namespace rsgpu

static void check_lists(const char *who, RSGPU_Postings *const *lists, size_t n_lists) {
  for (size_t l = 0; l < n_lists; l++) {
    if (!lists[l]) throw std::runtime_error(std::string(who) + ": NULL list");
    if (lists[l]->device != lists[0]->device) throw std::runtime_error(std::string(who) + ": lists on different devices");
  }
}

// helpers shared by RSGPU_Intersect(Ex) / RSGPU_Union / RSGPU_EvalTree and the fused query --------------------------------
// One child of the aggregate being built: a term's posting list, or the hit list of a nested union / intersection.
struct Source {
  const uint32_t *ids = nullptr;
  uint32_t len = 0;
  int n_leaves = 0;
  const uint32_t *freq[kMaxLists] = {};
  const uint32_t *epos[kMaxLists] = {};
  const RSGPU_Postings *src[kMaxLists] = {};
  int orig[kMaxLists] = {};  // the caller's list index of every leaf
  int op = 0;                // 0 term, 1 union, 2 intersection
  double weight = 1.0;
  uint64_t base = 0, first = 0, last = 0;  // ids[] are relative to base; first / last bound the doc ids from below / above
  std::vector<TNode> tree;   // the source's own result tree over its leaves (post-order, its root last)
};
// What an intersection sorts its children by (rqe_iterators/src/intersection.rs:94-119): num_estimated x
// intersection_sort_weight -- 1 / (number of children) for a child INTERSECTION (intersection.rs:580-582: fewer children, tighter
// selectivity), the number of children for a child UNION when the module's prioritizeIntersectUnionChildren is set (union_flat.rs:
// 817-823; off by default, src/config.h:451: knob prioritize_union_children), 1 for everything else.  op: 0 term, 1 union, 2 intersection.
static double intersection_sort_key(size_t estimate, int op, size_t n_children) {
  double w = 1.0;
  if (op == 2) w = 1.0 / (double)std::max<size_t>(n_children, 1);
  else if (op == 1 && scan_tuning().prioritize_union_children) w = (double)std::max<size_t>(n_children, 1);
  return (double)estimate * w;
}
static Source term_source(RSGPU_Postings *p, int orig) {
  Source s;
  s.ids = p->ids.p;
  s.len = p->n_entries;
  s.base = p->base;
  s.first = p->first_id;
  s.last = p->last;
  s.n_leaves = 1;
  s.freq[0] = p->cd.freq >= 0 ? p->freqs.p : nullptr;
  s.epos[0] = nullptr;  // the position IS the entry index
  s.src[0] = p;
  s.orig[0] = orig;
  s.tree.push_back(TNode{0, 0, 0, 1.0});
  return s;
}
static Source hits_source(const RSGPU_Hits *g, int op, double weight) {
  Source s;
  s.ids = g->ids.p;
  s.len = g->len;
  s.base = g->base;
  s.first = g->base;
  s.last = g->last;
  s.n_leaves = g->n_lists;
  for (int l = 0; l < g->n_lists; l++) {
    s.freq[l] = g->freqs.p + (size_t)l * g->cap;
    s.epos[l] = g->with_offsets ? g->epos.p + (size_t)l * g->cap : nullptr;
    s.src[l] = g->src[l];
    s.orig[l] = g->order[l];
  }
  s.op = op;
  s.weight = weight;
  s.tree = g->tree;
  if (!s.tree.empty()) {  // the nested list's root becomes this child: its operator and weight come from the query node
    s.tree.back().op = (uint8_t)op;
    s.tree.back().weight = weight;
  }
  return s;
}

static OffsetView offset_view(const RSGPU_Hits *h) {
  OffsetView o;
  memset(&o, 0, sizeof o);
  for (int s = 0; s < h->n_lists; s++) {
    const RSGPU_Postings *p = h->src[s];
    o.bytes[s] = p->bytes.p;
    o.off_pos[s] = p->has_offsets() ? p->off_pos.p : nullptr;
    o.off_len[s] = p->has_offsets() ? p->off_len.p : nullptr;
  }
  return o;
}
// proximity parameters of the hit list's own tree
static ProxParams tree_prox(const RSGPU_Hits *h, long max_slop, int in_order) {
  ProxParams P;
  memset(&P, 0, sizeof P);
  P.n_children = h->n_groups;
  P.n_leaves = h->n_lists;
  for (int g = 0; g <= h->n_groups; g++) P.child_first[g] = h->group_first[g];
  for (int g = 0; g < h->n_groups; g++) P.is_agg[g] = h->group_op[g] != 0;
  P.max_slop = max_slop < 0 ? -1 : (int)std::min<long>(max_slop, 0x7FFFFFFF);
  P.in_order = in_order;
  P.count_present = h->is_union ? 1 : 0;
  return P;
}

// Lays the sources out as the aggregate's children (ListView slot = group, leaves flattened in order) and records the
// tree shape in the hit list.  Returns the leaf map for the write / proximity kernels.
static LeafMap adopt_sources(RSGPU_Hits *h, const std::vector<Source> &srcs, ListView &v, int root_op) {
  memset(&v, 0, sizeof v);
  h->tree.clear();
  {
    int leaf0 = 0;
    for (const Source &s : srcs) {
      for (TNode t : s.tree) {
        if (t.op == 0) t.leaf = (uint8_t)(t.leaf + leaf0);
        h->tree.push_back(t);
      }
      leaf0 += s.n_leaves;
    }
    h->tree.push_back(TNode{(uint8_t)root_op, 0, (uint16_t)srcs.size(), 1.0});
    if (h->tree.size() > (size_t)kMaxNodes) throw std::runtime_error("query tree: more than 64 nodes");
  }
  LeafMap m;
  memset(&m, 0, sizeof m);
  v.n = (int)srcs.size();
  h->n_groups = (int)srcs.size();
  h->with_offsets = false;
  // the frame the children share: offsets from the smallest doc id among them; every child's ids must fit 32 bits
  // above it (a child's own base may lie below -- lists whose ids fit 32 bits keep base 0 -- or above it)
  uint64_t base = ~0ull, last = 0;
  for (const Source &s : srcs)
    if (s.len) {
      base = std::min(base, s.first);
      last = std::max(last, s.last);
    }
  if (base == ~0ull) base = 0;
  if (last >= base && last - base > 0xFFFFFFFEull)
    throw std::runtime_error("the posting lists of one query span 2^32 doc ids or more (device ids are 32-bit offsets from the smallest doc id)");
  h->base = base;
  h->last = last;
  int leaf = 0;
  for (size_t g = 0; g < srcs.size(); g++) {
    const Source &s = srcs[g];
    v.ids[g] = s.ids;
    v.len[g] = s.len;
    v.add[g] = s.len ? (long long)(s.base - base) : 0ll;  // (two's complement: negative when the child's base lies below)
    h->group_first[g] = (uint8_t)leaf;
    h->group_op[g] = (uint8_t)s.op;
    h->group_weight[g] = s.weight;
    for (int l = 0; l < s.n_leaves; l++, leaf++) {
      m.leaf_list[leaf] = (uint8_t)g;
      m.leaf_freq[leaf] = s.freq[l];
      m.leaf_epos[leaf] = s.epos[l];
      h->src[leaf] = s.src[l];
      h->order[leaf] = s.orig[l];
      h->with_offsets |= s.src[l]->has_offsets();
    }
  }
  h->group_first[srcs.size()] = (uint8_t)leaf;
  h->n_lists = m.n_leaves = leaf;
  return m;
}

constexpr uint32_t kCountPending = 0xFFFFFFFFu;  // (a hit count cannot reach it: n0 < 2^32 - 16)
// AND of the sources (already in iteration order; source 0 drives).  Enqueues probe [+ proximity filter] + scan + write
// on c->stream; *total_out (pinned host memory) receives the hit count once the stream has been synchronised.
static void combine_and(RSGPU_Hits *h, const std::vector<Source> &srcs, QueryCtx *c, Scratch &sc, uint32_t *total_out,
                        long max_slop, int in_order) {
  ListView v;
  const LeafMap m = adopt_sources(h, srcs, v, 2);
  h->is_union = false;
  const uint32_t n0 = v.len[0];
  h->cap = std::max<uint32_t>(n0, 1);
  h->ids.alloc(h->cap);
  h->freqs.alloc((size_t)h->cap * h->n_lists);
  if (h->with_offsets) h->epos.alloc((size_t)h->cap * h->n_lists);
  *total_out = 0;
  if (n0 == 0) return;
  *total_out = kCountPending;  // until scan_counts_kernel has written the real count (callers that poll instead of synchronising)
  // long driving lists without a proximity filter (whose kernel re-counts per 256 drivers): tiles of 1 024 drivers
  const bool prox = (max_slop >= 0 || in_order) && srcs.size() > 1 && h->with_offsets;
  const int dpt = (!prox && scan_tuning().probe_dpt == 4 && n0 >= (1u << 18)) ? 4 : 1;
  const uint32_t nb = (n0 + 256 * dpt - 1) / (256 * dpt);
  sc.flags.ensure(n0);
  sc.pos.ensure((size_t)n0 * std::max<size_t>(srcs.size() - 1, 1));
  sc.block_counts.ensure(nb);
  launch_intersect_probe(v, sc.flags.p, sc.pos.p, sc.block_counts.p, c->stream, dpt);
  if (prox) {
    // Intersection::current_is_relevant (intersection.rs:205-215): a consensus document outside the window is dropped
    launch_prox_filter(tree_prox(h, max_slop, in_order), offset_view(h), m, n0, sc.pos.p, sc.flags.p, sc.block_counts.p,
                       c->stream);
  }
  launch_scan_counts(sc.block_counts.p, nb, total_out, c->stream);  // total_out: pinned host memory
  launch_intersect_write(v, m, sc.flags.p, sc.pos.p, sc.block_counts.p, h->ids.p, h->freqs.p, h->cap, c->stream,
                         h->with_offsets ? h->epos.p : nullptr, dpt);
  HIP_CHECK(hipGetLastError());
}

// OR of the sources (union_flat.rs:223-257,297-320); synchronises the stream (the total comes from per-source counts)
static void combine_or(RSGPU_Hits *h, const std::vector<Source> &srcs, QueryCtx *c, Scratch &sc) {
  ListView v;
  const LeafMap m = adopt_sources(h, srcs, v, 1);
  h->is_union = true;
  const size_t n = srcs.size();
  size_t sum = 0, max_len = 0;
  for (size_t s = 0; s < n; s++) {
    sum += v.len[s];
    max_len = std::max<size_t>(max_len, v.len[s]);
  }
  if (sum > 0xFFFFFFF0ull) throw std::runtime_error("RSGPU_Union: more than 2^32 postings");
  h->cap = (uint32_t)std::max<size_t>(sum, 1);
  h->ids.alloc(h->cap);
  h->freqs.alloc((size_t)h->cap * h->n_lists);
  if (h->with_offsets) h->epos.alloc((size_t)h->cap * h->n_lists);
  h->len = 0;
  if (!sum) return;
  // flags of all sources back to back, prefix arrays [len+1] back to back, one block-count array per source
  sc.flags.ensure(sum);
  sc.pos.ensure(sum + n);
  const uint32_t nb_max = (uint32_t)((max_len + 256) / 256 + 1);
  sc.block_counts.ensure((size_t)nb_max * n);
  sc.total.ensure(n);
  UnionView u;
  memset(&u, 0, sizeof u);
  size_t foff = 0, poff = 0;
  std::vector<size_t> flag_at(n);
  for (size_t s = 0; s < n; s++) {
    const uint32_t len = v.len[s], nb = (len + 255) / 256;
    uint32_t *bc = sc.block_counts.p + s * nb_max;
    flag_at[s] = foff;
    u.prefix[s] = sc.pos.p + poff;
    if (len) {
      launch_union_flag(v, (int)s, sc.flags.p + foff, bc, c->stream);
      launch_scan_counts(bc, nb, sc.total.p + s, c->stream);
    } else {
      HIP_CHECK(hipMemsetAsync(sc.total.p + s, 0, sizeof(uint32_t), c->stream));
    }
    launch_union_prefix(sc.flags.p + foff, len, bc, sc.total.p + s, sc.pos.p + poff, c->stream);
    foff += len;
    poff += len + 1;
  }
  for (size_t s = 0; s < n; s++)
    if (v.len[s])
      launch_union_write(v, m, u, (int)s, sc.flags.p + flag_at[s], h->ids.p, h->freqs.p, h->cap, c->stream,
                         h->with_offsets ? h->epos.p : nullptr);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> totals(n);
  HIP_CHECK(hipMemcpyAsync(totals.data(), sc.total.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  uint64_t total = 0;
  for (uint32_t t : totals) total += t;
  h->len = (uint32_t)total;
}

// flat AND of posting lists: decode (cached) + combine_and over term sources
static void intersect_async(RSGPU_Hits *h, RSGPU_Postings *const *lists, size_t n_lists, QueryCtx *c, Scratch &sc,
                            uint32_t *total_out, long max_slop = -1, int in_order = 0) {
  std::vector<int> order(n_lists);
  std::iota(order.begin(), order.end(), 0);
  // children ordered by estimated size, ascending and stable (reference intersection.rs:94-119); in_order keeps the
  // caller's order -- it is the order the terms must appear in
  if (!in_order)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lists[a]->n_entries < lists[b]->n_entries; });
  for (size_t l = 0; l < n_lists; l++) {
    if (l + 1 < n_lists && decode_pair_on(lists[l], lists[l + 1], c)) l++;
    else decode_on(lists[l], c);
  }
  std::vector<Source> srcs;
  for (size_t s = 0; s < n_lists; s++) srcs.push_back(term_source(lists[order[s]], order[s]));
  combine_and(h, srcs, c, sc, total_out, max_slop, in_order);
}

// per-hit slop from the term offsets, once per hit list (a no-op for lists without offsets)
static const int32_t *hit_slops(RSGPU_Hits *h, QueryCtx *c) {
  if (!h->with_offsets || h->n_groups < 2 || !h->len) return nullptr;
  if (!h->slops_ready) {
    h->slops.ensure(h->cap);
    launch_prox_slop(tree_prox(h, -1, 0), offset_view(h), h->epos.p, h->len, h->cap, h->slops.p, c->stream);
    HIP_CHECK(hipGetLastError());
    h->slops_ready = true;
  }
  return h->slops.p;
}
static bool slop_dependent(int scorer) {
  return scorer == RSGPU_SCORER_TFIDF || scorer == RSGPU_SCORER_TFIDF_DOCNORM || scorer == RSGPU_SCORER_BM25;
}
static void tree_score_params(ScoreParams &P, const RSGPU_Hits *h, const RSGPU_DocTable *t) {
  // table entry of hit id x (relative to the hits' base): x + (hits base - table first); may be negative
  P.table_off = t ? (long long)(h->base - t->first) : 0;  // (t == NULL: the tree alone -- a query without a scoring branch)
  P.n_lists = h->n_lists;
  P.n_groups = h->n_groups;
  for (int g = 0; g <= h->n_groups; g++) P.group_first[g] = h->group_first[g];
  for (int g = 0; g < h->n_groups; g++) {
    P.group_op[g] = h->group_op[g];
    P.group_weight[g] = h->group_weight[g];
  }
  // IndexResult_MinOffsetDelta over children without offsets returns num-1 (reference
  // src/index_result/index_result.c:102), 1 for a single child
  P.slop = h->n_groups > 1 ? h->n_groups - 1 : 1;
  P.is_union = h->is_union ? 1 : 0;
  // deeper than root -> groups -> leaves: the generic post-order evaluation
  P.n_nodes = 0;
  const int n = (int)h->tree.size();
  if (n >= 2 && n <= kMaxNodes) {
    struct Frame {
      int remaining, depth, op;
    };
    std::vector<Frame> st;
    std::vector<int> depth(n, 0), in_union(n, 0);
    st.push_back(Frame{(int)h->tree[n - 1].n_children, 0, (int)h->tree[n - 1].op});
    int max_depth = 0;
    for (int i = n - 2; i >= 0; i--) {
      while (!st.empty() && st.back().remaining == 0) st.pop_back();
      if (st.empty()) throw std::runtime_error("query tree: malformed node array");
      Frame &par = st.back();
      depth[i] = par.depth + 1;
      in_union[i] = par.op == 1;
      par.remaining--;
      max_depth = std::max(max_depth, depth[i]);
      if (h->tree[i].op != 0) st.push_back(Frame{(int)h->tree[i].n_children, depth[i], (int)h->tree[i].op});
    }
    if (max_depth > kMaxTreeDepth) throw std::runtime_error("query tree: nested deeper than 16 levels");
    if (max_depth > 2) {
      P.n_nodes = n;
      for (int i = 0; i < n; i++) {
        P.node_op[i] = h->tree[i].op;
        P.node_depth[i] = (uint8_t)depth[i];
        P.node_leaf[i] = h->tree[i].leaf;
        P.node_in_union[i] = (uint8_t)in_union[i];
        P.node_weight[i] = h->tree[i].weight;
      }
    }
  }
}

extern "C" {

RSGPU_Postings *RSGPU_Postings_Upload(int codec, size_t n_blocks, const uint64_t *first_doc_id,
                                      const uint64_t *last_doc_id, const uint32_t *num_entries,
                                      const uint64_t *byte_offset, const uint8_t *bytes) {
  S_TRY
  CodecDesc cd = codec_desc(codec);
  if (cd.kind < 0) throw std::runtime_error("unknown posting codec");
  // the arguments describe memory this function reads: refuse what cannot be a block list before touching it
  if (n_blocks > 0xFFFFFFFEull) throw std::runtime_error("more than 2^32 - 2 blocks");
  if (n_blocks && (!first_doc_id || !last_doc_id || !num_entries || !byte_offset)) throw std::runtime_error("a block array is NULL");
  {
    uint64_t total = 0;
    for (size_t b = 0; b < n_blocks; b++) {
      if (byte_offset[b] > byte_offset[b + 1]) throw std::runtime_error("byte_offset is not ascending");
      total += num_entries[b];
    }
    if (total > 0xFFFFFFFEull) throw std::runtime_error("a posting list of 2^32 entries or more is not supported on the device path");
    if (n_blocks && byte_offset[n_blocks] && !bytes) throw std::runtime_error("the encoded bytes are NULL");
  }
  std::string why;
  if (!device_available(&why)) throw std::runtime_error(why);
  auto *p = new RSGPU_Postings();
  std::unique_ptr<RSGPU_Postings> guard(p);
  HIP_CHECK(hipGetDevice(&p->device));
  p->codec = codec;
  p->cd = cd;
  p->n_blocks = (uint32_t)n_blocks;
  std::vector<uint32_t> first(n_blocks), eoff(n_blocks + 1, 0);
  uint64_t lo = ~0ull, hi = 0;
  for (size_t b = 0; b < n_blocks; b++) {
    lo = std::min(lo, first_doc_id[b]);
    hi = std::max(hi, std::max(first_doc_id[b], last_doc_id[b]));
  }
  // t_docId is 64-bit; the device holds 32-bit offsets from a per-list base (0 while every id fits 32 bits)
  p->base = (n_blocks && hi > 0xFFFFFFFEull) ? lo : 0;
  p->first_id = n_blocks ? lo : 0;
  p->last = hi;
  if (n_blocks && hi - p->base > 0xFFFFFFFEull) throw std::runtime_error("a posting list spanning 2^32 doc ids or more is not supported on the device path");
  for (size_t b = 0; b < n_blocks; b++) {
    first[b] = (uint32_t)(first_doc_id[b] - p->base);
    eoff[b + 1] = eoff[b] + num_entries[b];
  }
  p->n_entries = eoff[n_blocks];
  p->n_bytes = n_blocks ? (size_t)byte_offset[n_blocks] : 0;
  p->bytes.alloc(p->n_bytes + 16);  // slack: a truncated record never reads past the allocation
  if (p->n_bytes) HIP_CHECK(hipMemcpy(p->bytes.p, bytes, p->n_bytes, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemset(p->bytes.p + p->n_bytes, 0, 16));
  p->byte_off.upload(byte_offset, n_blocks + 1);
  p->first.upload(first.data(), n_blocks);
  p->nent.upload(num_entries, n_blocks);
  p->entry_off.upload(eoff.data(), n_blocks + 1);
  p->ids.alloc(p->n_entries);
  p->freqs.alloc(p->n_entries);
  p->masks.alloc((cd.mask >= 0 || cd.wide) ? p->n_entries : 1);
  if (cd.wide) p->wmasks.alloc((size_t)p->n_entries * 4);
  // sub-block sync points of the qint layouts without inline offsets (8 bytes per 16 postings; written by the first decode)
  // -- for lists uploaded in decode-per-query mode: with the decoded arrays cached a list is decoded once
  // (round 4: lists with inline offsets -- Full, the *Offsets codecs -- too, as long as every 8-block span fits the decode
  // kernel's staging buffer: their blocks are 2-3 x as long, a lane per block was 0.1 TB/s of encoded bytes)
  if (decode_sync_supported(cd) && n_blocks && !scan_tuning().cache_decoded && scan_tuning().decode_sync) {
    const size_t bpw = decode_sync_blocks_per_wave();
    uint64_t widest = 0;
    for (size_t b = 0; b < n_blocks; b += bpw)
      widest = std::max<uint64_t>(widest, byte_offset[std::min(b + bpw, n_blocks)] - (byte_offset[b] & ~15ull));
    if (cd.osz < 0 || widest + 256 < decode_stage_bytes()) {
      p->sync.alloc(decode_sync_words((uint32_t)n_blocks));
      p->sync_span = (uint32_t)std::min<uint64_t>(widest, 0xFFFFFFFFull);
    }
  }
  if (cd.osz >= 0) {
    // the decoded offsets index addresses the byte buffer with 32 bits
    if (p->n_bytes >= 0xFFFFFFF0ull) throw std::runtime_error("posting lists with offsets are limited to 4 GiB of encoded bytes");
    p->off_pos.alloc(p->n_entries);
    p->off_len.alloc(p->n_entries);
  }
  return guard.release();
  S_CATCH(nullptr)
}
void RSGPU_Postings_Free(RSGPU_Postings *p) { delete p; }
size_t RSGPU_Postings_NumEntries(const RSGPU_Postings *p) { return p ? p->n_entries : 0; }
size_t RSGPU_Postings_NumBytes(const RSGPU_Postings *p) { return p ? p->n_bytes : 0; }

long RSGPU_Postings_Decode(RSGPU_Postings *p, uint64_t *doc_ids_out, uint32_t *freqs_out, uint32_t *masks_out) {
  if (!p) return -1;
  S_TRY
  HIP_CHECK(hipSetDevice(p->device));
  CtxLease c(p->device);
  StageTimer t(c.c, 0);
  // an explicit decode request always runs the kernel (it is how the decode stage is measured)
  decode_on(p, c.c, true);
  HIP_CHECK(hipStreamSynchronize(c->stream));
  p->decoded.store(true, std::memory_order_release);
  t.stop();
  const uint32_t n = p->n_entries;
  if (doc_ids_out && n) {
    std::vector<uint32_t> tmp(n);
    HIP_CHECK(hipMemcpy(tmp.data(), p->ids.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) doc_ids_out[i] = p->base + tmp[i];
  }
  if (freqs_out && n) {
    if (p->cd.freq >= 0) HIP_CHECK(hipMemcpy(freqs_out, p->freqs.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    else std::fill(freqs_out, freqs_out + n, 1u);  // the term record's default (index_result/src/core/mod.rs:192-197)
  }
  if (masks_out && n) {
    if (p->cd.mask >= 0 || p->cd.wide) HIP_CHECK(hipMemcpy(masks_out, p->masks.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    else memset(masks_out, 0, n * sizeof(uint32_t));
  }
  return (long)n;
  S_CATCH(-1)
}

long RSGPU_Postings_DecodeWideMasks(RSGPU_Postings *p, uint64_t *masks_lo_out, uint64_t *masks_hi_out) {
  if (!p) return -1;
  S_TRY
  if (!p->cd.wide) throw std::runtime_error("RSGPU_Postings_DecodeWideMasks: not a wide codec");
  if (RSGPU_Postings_Decode(p, nullptr, nullptr, nullptr) < 0) return -1;
  HIP_CHECK(hipSetDevice(p->device));
  const uint32_t n = p->n_entries;
  std::vector<uint32_t> w((size_t)n * 4);
  if (n) HIP_CHECK(hipMemcpy(w.data(), p->wmasks.p, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; i++) {
    if (masks_lo_out) masks_lo_out[i] = (uint64_t)w[4 * (size_t)i] | ((uint64_t)w[4 * (size_t)i + 1] << 32);
    if (masks_hi_out) masks_hi_out[i] = (uint64_t)w[4 * (size_t)i + 2] | ((uint64_t)w[4 * (size_t)i + 3] << 32);
  }
  return (long)n;
  S_CATCH(-1)
}

// with a window (max_slop / in_order) through the general hybrid tile kernel (defined next to it, below): NULL = stage by stage
static RSGPU_Hits *intersect_tiles(RSGPU_Postings *const *lists, size_t n_lists, long max_slop, int in_order);
RSGPU_Hits *RSGPU_IntersectEx(RSGPU_Postings *const *lists, size_t n_lists, long max_slop, int in_order) {
  if (!lists || !n_lists || n_lists > (size_t)kMaxLists) {
    last_error() = "RSGPU_Intersect: 1..32 lists";
    return nullptr;
  }
  S_TRY
  check_lists("RSGPU_Intersect", lists, n_lists);
  const int device = lists[0]->device;
  HIP_CHECK(hipSetDevice(device));
  // A phrase / proximity intersection (max_slop / in_order over lists that store offsets): the general hybrid tile kernel tests
  // the candidates' windows where it finds them and writes the hit list (round 4) -- the staged form's prox_filter_kernel is a
  // launch of its own over every driver, between the probe and the ordered write.  Same hit list.
  if (RSGPU_Hits *fast = intersect_tiles(lists, n_lists, max_slop, in_order)) return fast;
  CtxLease c(device);
  auto *h = new RSGPU_Hits();
  std::unique_ptr<RSGPU_Hits> guard(h);
  h->device = device;
  h->n_lists = (int)n_lists;
  Scratch &sc = scratch(device);
  const bool prof = scan_profile().enabled.load(std::memory_order_relaxed) != 0;
  if (prof) {  // the decode stage on its own (a stream sync), as the per-stage profile reports it
    StageTimer td(c.c, 0);
    for (size_t l = 0; l < n_lists; l++) decode_on(lists[l], c.c);
    td.stop();
  }
  StageTimer ti(c.c, 1);
  intersect_async(h, lists, n_lists, c.c, sc, c->h_counters, max_slop, in_order);
  ti.stop();  // synchronises the stream: the count below is final
  h->len = c->h_counters[0];
  return guard.release();
  S_CATCH(nullptr)
}
RSGPU_Hits *RSGPU_Intersect(RSGPU_Postings *const *lists, size_t n_lists) {
  return RSGPU_IntersectEx(lists, n_lists, -1, 0);
}
// reference src/redisearch_rs/rqe_iterators/src/union_flat.rs:223-257,297-320
RSGPU_Hits *RSGPU_Union(RSGPU_Postings *const *lists, size_t n_lists) {
  if (!lists || !n_lists || n_lists > (size_t)kMaxLists) {
    last_error() = "RSGPU_Union: 1..32 lists";
    return nullptr;
  }
  S_TRY
  check_lists("RSGPU_Union", lists, n_lists);
  const int device = lists[0]->device;
  HIP_CHECK(hipSetDevice(device));
  CtxLease c(device);
  auto *h = new RSGPU_Hits();
  std::unique_ptr<RSGPU_Hits> guard(h);
  h->device = device;
  StageTimer td(c.c, 0);
  for (size_t l = 0; l < n_lists; l++) decode_on(lists[l], c.c);
  td.stop();
  std::vector<Source> srcs;  // caller order: a union has no driving child
  for (size_t s = 0; s < n_lists; s++) srcs.push_back(term_source(lists[s], (int)s));
  StageTimer ti(c.c, 1);
  combine_or(h, srcs, c.c, scratch(device));
  ti.stop();
  return guard.release();
  S_CATCH(nullptr)
}

// RSGPU_EvalTree through the general hybrid tile kernel (defined next to it, below): NULL = not such a tree, stage by stage
static RSGPU_Hits *eval_tree_tiles(const RSGPU_TreeQuery *q, size_t n_lists);

/* Two-level query tree: root (AND / OR) over groups, each a term or an OR / AND of terms -- e.g. the stemmer's
 * (run|running|ran) (shoe|shoes), or (a b) | (c d).  Nested groups are evaluated first (their hit lists stay alive
 * inside the result), the root then combines the groups' id lists and carries every term's frequency and entry index
 * along, so that scoring sees the reference's result tree (Intersection{Union{..},Union{..}} etc.). */
RSGPU_Hits *RSGPU_EvalTree(const RSGPU_TreeQuery *q) {
  if (!q || !q->lists || !q->n_groups || !q->group_first) {
    last_error() = "RSGPU_EvalTree: empty tree";
    return nullptr;
  }
  S_TRY
  const size_t n_lists = q->group_first[q->n_groups];
  if (q->n_groups > (size_t)kMaxLists || n_lists > (size_t)kMaxLists || !n_lists)
    throw std::runtime_error("RSGPU_EvalTree: at most 32 groups and 32 terms");
  if (q->root_op != RSGPU_OP_INTERSECT && q->root_op != RSGPU_OP_UNION) throw std::runtime_error("RSGPU_EvalTree: bad root_op");
  check_lists("RSGPU_EvalTree", q->lists, n_lists);
  const int device = q->lists[0]->device;
  HIP_CHECK(hipSetDevice(device));
  // A root intersection with an aggregate child (a (b|c), (a b) c ...) over at most eight lists: the general hybrid tile kernel
  // probes every list in place and writes the hit list (round 4) -- no child is materialised first.  Same hit list.
  if (RSGPU_Hits *fast = eval_tree_tiles(q, n_lists)) return fast;
  CtxLease c(device);
  Scratch &sc = scratch(device);
  auto *h = new RSGPU_Hits();
  std::unique_ptr<RSGPU_Hits> guard(h);
  h->device = device;
  for (size_t l = 0; l < n_lists; l++) decode_on(q->lists[l], c.c);
  struct Grp {
    Source s;
    size_t estimate;
    int index;
    double key = 0.0;  // what the root intersection sorts by
  };
  std::vector<Grp> groups;
  for (size_t g = 0; g < q->n_groups; g++) {
    const size_t a = q->group_first[g], b = q->group_first[g + 1];
    if (b <= a || b > n_lists) throw std::runtime_error("RSGPU_EvalTree: bad group_first");
    const int op = q->group_op ? q->group_op[g] : RSGPU_OP_TERM;
    const double w = q->group_weight ? q->group_weight[g] : 1.0;
    if (op == RSGPU_OP_TERM || b - a == 1) {
      if (b - a != 1) throw std::runtime_error("RSGPU_EvalTree: a term group holds exactly one list");
      if (op == RSGPU_OP_TERM) {
        groups.push_back(Grp{term_source(q->lists[a], (int)a), q->lists[a]->n_entries, (int)g,
                             intersection_sort_key(q->lists[a]->n_entries, 0, 1)});
        continue;
      }
    }
    // nested aggregate: its own hit list first
    std::unique_ptr<RSGPU_Hits> sub(new RSGPU_Hits());
    sub->device = device;
    std::vector<int> order(b - a);
    std::iota(order.begin(), order.end(), (int)a);
    if (op == RSGPU_OP_INTERSECT)
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return q->lists[x]->n_entries < q->lists[y]->n_entries; });
    std::vector<Source> srcs;
    for (int li : order) srcs.push_back(term_source(q->lists[li], li));
    size_t est = 0;
    if (op == RSGPU_OP_INTERSECT) {
      combine_and(sub.get(), srcs, c.c, sc, c->h_counters, -1, 0);
      HIP_CHECK(hipStreamSynchronize(c->stream));
      sub->len = c->h_counters[0];
      est = q->lists[order[0]]->n_entries;  // num_estimated of an intersection: its smallest child
    } else if (op == RSGPU_OP_UNION) {
      combine_or(sub.get(), srcs, c.c, sc);
      for (int li : order) est += q->lists[li]->n_entries;  // ... of a union: the sum
    } else {
      throw std::runtime_error("RSGPU_EvalTree: bad group_op");
    }
    groups.push_back(Grp{hits_source(sub.get(), op == RSGPU_OP_UNION ? 1 : 2, w), est, (int)g,
                         intersection_sort_key(est, op == RSGPU_OP_UNION ? 1 : 2, b - a)});
    h->nested.push_back(std::move(sub));
  }
  std::vector<Source> srcs;
  if (q->root_op == RSGPU_OP_INTERSECT) {
    // children sorted by estimate, ascending and stable, unless in_order pins the caller's order
    if (!q->in_order) std::stable_sort(groups.begin(), groups.end(), [](const Grp &x, const Grp &y) { return x.key < y.key; });
    for (auto &g : groups) srcs.push_back(g.s);
    combine_and(h, srcs, c.c, sc, c->h_counters, q->max_slop, q->in_order);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    h->len = c->h_counters[0];
  } else {
    for (auto &g : groups) srcs.push_back(g.s);
    combine_or(h, srcs, c.c, sc);
  }
  return guard.release();
  S_CATCH(nullptr)
}

/* Query trees of any depth (include/rsgpu_search.h RSGPU_EvalTreeNodes): the node array is walked once, bottom-up; every
 * aggregate becomes a hit list of its own (kept alive inside the result) whose term columns and result tree travel up
 * with it, exactly as the reference nests iterators (each Intersection / Union iterator owns its children and yields an
 * aggregate result holding theirs: rqe_iterators/src/intersection.rs:313-339, union_flat.rs:297-320). */
RSGPU_Hits *RSGPU_EvalTreeNodes(const RSGPU_TreeNode *nodes, size_t n_nodes, RSGPU_Postings *const *lists, size_t n_lists) {
  if (!nodes || !n_nodes || !lists || !n_lists) {
    last_error() = "RSGPU_EvalTreeNodes: empty tree";
    return nullptr;
  }
  S_TRY
  if (n_nodes > (size_t)kMaxNodes || n_lists > (size_t)kMaxLists) throw std::runtime_error("RSGPU_EvalTreeNodes: at most 64 nodes over 32 terms");
  check_lists("RSGPU_EvalTreeNodes", lists, n_lists);
  const int device = lists[0]->device;
  HIP_CHECK(hipSetDevice(device));
  CtxLease c(device);
  Scratch &sc = scratch(device);
  struct Sub {
    Source s;
    size_t estimate = 0;
    double key = 0.0;                 // estimate x intersection_sort_weight: what a parent intersection sorts by
    std::unique_ptr<RSGPU_Hits> own;  // an aggregate's own hit list (NULL for a term)
  };
  std::vector<Sub> st;
  std::vector<char> used(n_lists, 0);
  size_t leaves = 0;
  for (size_t i = 0; i < n_nodes; i++) {
    const RSGPU_TreeNode &nd = nodes[i];
    const bool root = i + 1 == n_nodes;
    if (nd.op == RSGPU_OP_TERM) {
      if (nd.list >= n_lists) throw std::runtime_error("RSGPU_EvalTreeNodes: term node names a list that is not there");
      if (used[nd.list]) throw std::runtime_error("RSGPU_EvalTreeNodes: a list may appear once in the tree");
      used[nd.list] = 1;
      if (++leaves > (size_t)kMaxLists) throw std::runtime_error("RSGPU_EvalTreeNodes: more than 32 terms");
      decode_on(lists[nd.list], c.c);
      Sub t;
      t.s = term_source(lists[nd.list], (int)nd.list);
      t.estimate = lists[nd.list]->n_entries;
      t.key = intersection_sort_key(t.estimate, 0, 1);
      if (!root) {
        st.push_back(std::move(t));
        continue;
      }
      // a bare term as the whole query: the intersection of one list
      if (!st.empty()) throw std::runtime_error("RSGPU_EvalTreeNodes: subtrees left over (the last node must be the root)");
      std::unique_ptr<RSGPU_Hits> h(new RSGPU_Hits());
      h->device = device;
      combine_and(h.get(), {t.s}, c.c, sc, c->h_counters, -1, 0);
      HIP_CHECK(hipStreamSynchronize(c->stream));
      h->len = c->h_counters[0];
      return h.release();
    }
    if (nd.op != RSGPU_OP_UNION && nd.op != RSGPU_OP_INTERSECT) throw std::runtime_error("RSGPU_EvalTreeNodes: bad op");
    if (!nd.n_children || nd.n_children > st.size() || nd.n_children > (size_t)kMaxLists)
      throw std::runtime_error("RSGPU_EvalTreeNodes: an aggregate needs 1..32 complete subtrees before it (post-order)");
    std::vector<Sub> kids;
    for (size_t k = st.size() - nd.n_children; k < st.size(); k++) kids.push_back(std::move(st[k]));
    st.resize(st.size() - nd.n_children);
    // an intersection iterates its children by ascending estimate, stable, unless in_order pins the query's order
    // (intersection.rs:94-119); a union keeps the query's order
    if (nd.op == RSGPU_OP_INTERSECT && !nd.in_order)
      std::stable_sort(kids.begin(), kids.end(), [](const Sub &x, const Sub &y) { return x.key < y.key; });
    std::vector<Source> srcs;
    for (Sub &k : kids) srcs.push_back(k.s);
    std::unique_ptr<RSGPU_Hits> h(new RSGPU_Hits());
    h->device = device;
    size_t est = 0;
    if (nd.op == RSGPU_OP_INTERSECT) {
      combine_and(h.get(), srcs, c.c, sc, c->h_counters, nd.max_slop, nd.in_order);
      HIP_CHECK(hipStreamSynchronize(c->stream));
      h->len = c->h_counters[0];
      est = kids[0].estimate;
      for (const Sub &k : kids) est = std::min(est, k.estimate);
    } else {
      combine_or(h.get(), srcs, c.c, sc);
      for (const Sub &k : kids) est += k.estimate;
    }
    for (Sub &k : kids)
      if (k.own) h->nested.push_back(std::move(k.own));
    if (root) {
      if (!st.empty()) throw std::runtime_error("RSGPU_EvalTreeNodes: subtrees left over (the last node must be the root)");
      return h.release();
    }
    Sub up;
    up.s = hits_source(h.get(), nd.op == RSGPU_OP_UNION ? 1 : 2, nd.weight);
    up.estimate = est;
    up.key = intersection_sort_key(est, nd.op == RSGPU_OP_UNION ? 1 : 2, nd.n_children);
    up.own = std::move(h);
    st.push_back(std::move(up));
  }
  throw std::runtime_error("RSGPU_EvalTreeNodes: the node array does not end in a root");
  S_CATCH(nullptr)
}

/* the result tree of a hit list, post-order (RSGPU_EvalTreeNodes' own order after the intersections sorted their children):
 * op / leaf column / children / weight per node; any output may be NULL.  Returns the number of nodes or -1. */
int RSGPU_Hits_TreeNodes(const RSGPU_Hits *h, int *op, int *leaf, int *n_children, double *weight) {
  if (!h) return -1;
  for (size_t i = 0; i < h->tree.size(); i++) {
    if (op) op[i] = h->tree[i].op;
    if (leaf) leaf[i] = h->tree[i].op == 0 ? h->tree[i].leaf : -1;
    if (n_children) n_children[i] = h->tree[i].n_children;
    if (weight) weight[i] = h->tree[i].weight;
  }
  return (int)h->tree.size();
}

// reference src/redisearch_rs/rqe_iterators/src/not.rs:171-209 (1..=max_doc_id) and not_optimized.rs (universe)
RSGPU_Hits *RSGPU_Not(RSGPU_Postings *child, RSGPU_Postings *universe, uint64_t max_doc_id) {
  if (!child) {
    last_error() = "RSGPU_Not: NULL child";
    return nullptr;
  }
  S_TRY
  if (universe && universe->device != child->device) throw std::runtime_error("RSGPU_Not: lists on different devices");
  // one frame for the child, the universe and max_doc_id: the universe's base (the plain NOT enumerates 1..max_doc_id
  // and needs all of it below 2^32)
  const uint64_t frame = universe ? universe->base : 0;
  if (!universe && child->n_entries && child->base != 0)
    throw std::runtime_error("RSGPU_Not: 1..max_doc_id must stay below 2^32 without a universe list");
  if (max_doc_id < frame) max_doc_id = frame;  // nothing qualifies
  max_doc_id -= frame;
  if (max_doc_id > 0xFFFFFFF0ull) {
    if (!universe) throw std::runtime_error("RSGPU_Not: 1..max_doc_id must stay below 2^32 without a universe list");
    max_doc_id = 0xFFFFFFF0ull;  // every id of the frame
  }
  const int device = child->device;
  HIP_CHECK(hipSetDevice(device));
  CtxLease c(device);
  auto *h = new RSGPU_Hits();
  std::unique_ptr<RSGPU_Hits> guard(h);
  h->device = device;
  h->n_lists = 1;  // one virtual child: freq 1; the caller scores it with idf = 1 (src/ext/default.c:289-293)
  h->order[0] = 0;
  h->n_groups = 1;
  h->group_first[0] = 0;
  h->group_first[1] = 1;
  h->group_op[0] = 0;
  h->group_weight[0] = 1.0;
  h->src[0] = child;
  h->base = frame;
  h->last = frame + max_doc_id;
  StageTimer td(c.c, 0);
  decode_on(child, c.c);
  if (universe) decode_on(universe, c.c);
  td.stop();
  const uint32_t max_doc = (uint32_t)max_doc_id;
  const uint32_t n_cand = universe ? universe->n_entries : max_doc;
  h->cap = std::max<uint32_t>(n_cand, 1);
  h->ids.alloc(h->cap);
  h->freqs.alloc(h->cap);
  if (!n_cand) return guard.release();
  Scratch &sc = scratch(device);
  StageTimer ti(c.c, 1);
  uint32_t total = 0;
  if (!universe) {
    // survivors = max_doc - (child entries <= max_doc); slots are computed per document, no scan needed
    launch_not_range(child->ids.p, child->n_entries, max_doc, h->ids.p, h->freqs.p, h->cap, c->stream);
    HIP_CHECK(hipGetLastError());
    ti.stop();
    sc.total.ensure(1);
    launch_count_below(child->ids.p, child->n_entries, max_doc_id + 1, sc.total.p, c->stream);
    uint32_t below = 0;
    HIP_CHECK(hipMemcpyAsync(&below, sc.total.p, sizeof below, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));  // (the query stream does not synchronise with the null stream)
    total = max_doc - below;
  } else {
    const uint32_t nb = (n_cand + 255) / 256;
    sc.flags.ensure(n_cand);
    sc.block_counts.ensure(nb);
    sc.total.ensure(1);
    launch_not_universe_flag(universe->ids.p, n_cand, child->ids.p, child->n_entries, (long long)(frame - child->base), max_doc, sc.flags.p,
                             sc.block_counts.p, c->stream);
    launch_scan_counts(sc.block_counts.p, nb, sc.total.p, c->stream);
    launch_not_universe_write(universe->ids.p, n_cand, sc.flags.p, sc.block_counts.p, h->ids.p, h->freqs.p, h->cap,
                              c->stream);
    HIP_CHECK(hipGetLastError());
    ti.stop();
    HIP_CHECK(hipMemcpy(&total, sc.total.p, sizeof total, hipMemcpyDeviceToHost));
  }
  h->len = total;
  return guard.release();
  S_CATCH(nullptr)
}
void RSGPU_Hits_Free(RSGPU_Hits *h) { delete h; }
size_t RSGPU_Hits_Len(const RSGPU_Hits *h) { return h ? h->len : 0; }

int RSGPU_Hits_Read(const RSGPU_Hits *hc, uint64_t *doc_ids, uint32_t *freqs) {
  if (!hc) return -1;
  RSGPU_Hits *h = const_cast<RSGPU_Hits *>(hc);
  S_TRY
  HIP_CHECK(hipSetDevice(h->device));
  if (doc_ids) {
    const std::vector<uint32_t> &ids = h->host_ids();
    for (uint32_t i = 0; i < h->len; i++) doc_ids[i] = h->base + ids[i];
  }
  if (freqs && h->len)
    for (int s = 0; s < h->n_lists; s++)  // back to the caller's list order
      HIP_CHECK(hipMemcpy(freqs + (size_t)h->order[s] * h->len, h->freqs.p + (size_t)s * h->cap,
                          h->len * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return 0;
  S_CATCH(-1)
}

// ---- the iterator seam's view of a hit list (include/rsgpu_search.h "record access"; query_iterators.c) -----------------
int RSGPU_Postings_Codec(const RSGPU_Postings *p) { return p ? p->codec : -1; }
int RSGPU_Hits_IsUnion(const RSGPU_Hits *h) { return h && h->is_union ? 1 : 0; }
size_t RSGPU_Hits_NumLeaves(const RSGPU_Hits *h) { return h ? (size_t)h->n_lists : 0; }

int RSGPU_Hits_Tree(const RSGPU_Hits *h, int *root_is_union, int *group_first, int *group_op, double *group_weight) {
  if (!h) return -1;
  if (root_is_union) *root_is_union = h->is_union ? 1 : 0;
  for (int g = 0; g < h->n_groups; g++) {
    if (group_first) group_first[g] = h->group_first[g];
    if (group_op) group_op[g] = h->group_op[g];
    if (group_weight) group_weight[g] = h->group_weight[g];
  }
  if (group_first) group_first[h->n_groups] = h->group_first[h->n_groups];
  return h->n_groups;
}

int RSGPU_Hits_LeafOrder(const RSGPU_Hits *h, int *list_of_child) {
  if (!h || !list_of_child) return -1;
  for (int s = 0; s < h->n_lists; s++) list_of_child[s] = h->order[s];
  return h->n_lists;
}

long RSGPU_Hits_ReadRange(const RSGPU_Hits *hc, size_t first, size_t count, uint64_t *doc_ids) {
  if (!hc || !doc_ids) return -1;
  RSGPU_Hits *h = const_cast<RSGPU_Hits *>(hc);
  S_TRY
  if (first >= h->len) return 0;
  count = std::min<size_t>(count, h->len - first);
  HIP_CHECK(hipSetDevice(h->device));
  std::vector<uint32_t> tmp(count);
  if (count) HIP_CHECK(hipMemcpy(tmp.data(), h->ids.p + first, count * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < count; i++) doc_ids[i] = h->base + tmp[i];
  return (long)count;
  S_CATCH(-1)
}

long RSGPU_Hits_ReadRecords(const RSGPU_Hits *hc, size_t list, size_t first, size_t count, uint32_t *entry, uint32_t *freqs,
                            uint64_t *mask_lo, uint64_t *mask_hi, uint64_t *offsets_pos, uint32_t *offsets_len) {
  if (!hc) return -1;
  RSGPU_Hits *h = const_cast<RSGPU_Hits *>(hc);
  S_TRY
  int slot = -1;
  for (int s = 0; s < h->n_lists; s++)
    if ((size_t)h->order[s] == list) slot = s;
  if (slot < 0) throw std::runtime_error("RSGPU_Hits_ReadRecords: no such list in this hit list");
  if (first >= h->len) return 0;
  count = std::min<size_t>(count, h->len - first);
  if (!count) return 0;
  RSGPU_Postings *p = const_cast<RSGPU_Postings *>(h->src[slot]);
  if (!p) throw std::runtime_error("RSGPU_Hits_ReadRecords: the hit list does not know its posting lists");
  HIP_CHECK(hipSetDevice(h->device));
  CtxLease c(h->device);
  decode_on(p, c.c);
  DevBuf<uint32_t> out;
  out.alloc(7 * count);
  // the hit's id in the list's frame = hits base + id - list base
  const long long shift = (long long)(h->base - p->base);
  launch_hit_records(h->ids.p, (uint32_t)first, (uint32_t)count,
                     h->with_offsets && h->epos.p ? h->epos.p + (size_t)slot * h->cap : nullptr, p->ids.p, p->n_entries, shift,
                     (p->cd.mask >= 0 || p->cd.wide) ? p->masks.p : nullptr, p->cd.wide ? p->wmasks.p : nullptr,
                     p->has_offsets() ? p->off_pos.p : nullptr, p->has_offsets() ? p->off_len.p : nullptr, out.p, c->stream);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> host(7 * count), hf;
  HIP_CHECK(hipMemcpyAsync(host.data(), out.p, host.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  // (always: a hit's frequency column also says whether the leaf is part of the match -- 0 = a union child, or a
  // nested group, that did not match this document; every record that matched carries a frequency >= 1)
  hf.resize(count);
  HIP_CHECK(hipMemcpyAsync(hf.data(), h->freqs.p + (size_t)slot * h->cap + first, count * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < count; i++) {
    const uint32_t e = hf[i] == 0 ? 0xFFFFFFFFu : host[i];
    if (entry) entry[i] = e;
    if (freqs) freqs[i] = e == 0xFFFFFFFFu ? 0 : hf[i];
    if (mask_lo) mask_lo[i] = (uint64_t)host[count + i] | ((uint64_t)host[2 * count + i] << 32);
    if (mask_hi) mask_hi[i] = (uint64_t)host[3 * count + i] | ((uint64_t)host[4 * count + i] << 32);
    if (offsets_pos) offsets_pos[i] = host[5 * count + i];
    if (offsets_len) offsets_len[i] = host[6 * count + i];
  }
  return (long)count;
  S_CATCH(-1)
}

int RSGPU_Postings_ReadBytes(const RSGPU_Postings *p, size_t pos, size_t len, uint8_t *out) {
  if (!p || (!out && len)) return -1;
  S_TRY
  if (pos > p->n_bytes || len > p->n_bytes - pos) throw std::runtime_error("RSGPU_Postings_ReadBytes: range outside the list's bytes");
  HIP_CHECK(hipSetDevice(p->device));
  if (len) HIP_CHECK(hipMemcpy(out, p->bytes.p + pos, len, hipMemcpyDeviceToHost));
  return 0;
  S_CATCH(-1)
}

RSGPU_DocTable *RSGPU_DocTable_UploadWindow(uint64_t first_doc_id, size_t n, const uint32_t *doc_len,
                                            const float *doc_score, const uint32_t *max_term_freq);
RSGPU_DocTable *RSGPU_DocTable_Upload(size_t n, const uint32_t *doc_len, const float *doc_score,
                                      const uint32_t *max_term_freq) {
  return RSGPU_DocTable_UploadWindow(0, n, doc_len, doc_score, max_term_freq);
}
RSGPU_DocTable *RSGPU_DocTable_UploadWindow(uint64_t first_doc_id, size_t n, const uint32_t *doc_len,
                                            const float *doc_score, const uint32_t *max_term_freq) {
  S_TRY
  if (n > 0xFFFFFFF0ull) throw std::runtime_error("RSGPU_DocTable_Upload: at most 2^32 entries");
  std::string why;
  if (!device_available(&why)) throw std::runtime_error(why);
  auto *t = new RSGPU_DocTable();
  std::unique_ptr<RSGPU_DocTable> guard(t);
  HIP_CHECK(hipGetDevice(&t->device));
  t->n = (uint32_t)n;
  t->first = first_doc_id;
  t->doc_len.upload(doc_len, n);
  t->doc_score.upload(doc_score, n);
  if (max_term_freq) t->max_freq.upload(max_term_freq, n);
  if (n && scan_tuning().hybrid_packed_docs) {
    t->len_score.alloc(n);
    launch_pack_len_score(t->doc_len.p, t->doc_score.p, (uint32_t)n, t->len_score.p, nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(nullptr));
  }
  return guard.release();
  S_CATCH(nullptr)
}
void RSGPU_DocTable_Free(RSGPU_DocTable *t) { delete t; }

int RSGPU_Hits_Score(RSGPU_Hits *h, const RSGPU_DocTable *t, const RSGPU_ScoreArgs *a, double *scores_out) {
  if (!h || !t || !a) return -1;
  S_TRY
  if (t->device != h->device) throw std::runtime_error("RSGPU_Hits_Score: hits and document table live on different devices");
  HIP_CHECK(hipSetDevice(h->device));
  CtxLease c(h->device);
  ScoreParams P;
  memset(&P, 0, sizeof P);
  // BM25STD.NORM = BM25STD, then every score divided by the largest one (RPMaxScoreNormalizer sits right behind the
  // scorer in the reference pipeline, src/pipeline/pipeline_construction.c:546-547)
  const bool max_norm = a->scorer == RSGPU_SCORER_BM25STD_NORM;
  P.scorer = max_norm ? (int)RSGPU_SCORER_BM25STD : a->scorer;
  tree_score_params(P, h, t);
  P.avg_doc_len = a->avg_doc_len;
  P.root_weight = a->root_weight;
  P.min_score = a->min_score;
  P.inv_tanh = a->tanh_factor ? 1 / (double)a->tanh_factor : 0.0;
  for (int s = 0; s < h->n_lists; s++) {
    int o = h->order[s];
    P.idf[s] = a->idf ? a->idf[o] : 0.0;
    P.bm25_idf[s] = a->bm25_idf ? a->bm25_idf[o] : 0.0;
    P.weight[s] = a->weight ? a->weight[o] : 1.0;
  }
  h->scores.ensure(h->cap);
  h->keys.ensure(h->cap);
  if (slop_dependent(P.scorer)) P.slops = hit_slops(h, c.c);  // IndexResult_MinOffsetDelta from the term offsets
  StageTimer ts(c.c, 2);
  launch_score(P, h->ids.p, h->freqs.p, h->len, h->cap, t->doc_len.p, t->doc_score.p,
               t->max_freq.p, t->n, h->scores.p, h->keys.p, c->stream);
  if (max_norm && h->len) {
    Scratch &sc = scratch(h->device);
    sc.maxkey.ensure(1);
    HIP_CHECK(hipMemsetAsync(sc.maxkey.p, 0, sizeof(uint64_t), c->stream));
    launch_score_max_normalize(h->scores.p, h->keys.p, h->len, sc.maxkey.p, c->stream);
  }
  HIP_CHECK(hipGetLastError());
  ts.stop();
  h->scored = true;
  if (scores_out && h->len) HIP_CHECK(hipMemcpy(scores_out, h->scores.p, h->len * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
  S_CATCH(-1)
}

long RSGPU_Hits_TopN(RSGPU_Hits *h, size_t n, uint64_t *doc_ids_out, double *scores_out) {
  if (!h || !h->scored) {
    last_error() = "RSGPU_Hits_TopN: score the hits first";
    return -1;
  }
  S_TRY
  HIP_CHECK(hipSetDevice(h->device));
  CtxLease c(h->device);
  uint32_t k = (uint32_t)std::min<size_t>(n, h->len);
  std::vector<Hit> hits;
  StageTimer tt(c.c, 3);
  radix_select(c.c, h->keys.p, 8, h->len, k, Bound(), hits, nullptr);
  tt.stop();
  const std::vector<uint32_t> &ids = h->host_ids();
  for (size_t i = 0; i < hits.size(); i++) {
    if (doc_ids_out) doc_ids_out[i] = h->base + ids[hits[i].row];
    if (scores_out) scores_out[i] = key2score(hits[i].key);
  }
  return (long)hits.size();
  S_CATCH(-1)
}

long RSGPU_Hits_KnnRerank(RSGPU_Hits *h, VecSimIndex *index, const void *query, size_t k, uint64_t *doc_ids_out,
                          double *dist_out) {
  if (!h || !index || !query) return -1;
  FlatIndex *f = index->flat;
  S_TRY
  if (!f) {
    // a sharded handle (VecSimIndex_New under the "shards" knob): the candidates go through the ad-hoc context of the
    // VecSim ABI, which routes every label to the shard that owns it
    if (!h->len || !k) return 0;
    HIP_CHECK(hipSetDevice(h->device));
    const std::vector<uint32_t> &ids = h->host_ids();
    std::vector<size_t> labels(ids.size());
    for (size_t i = 0; i < ids.size(); i++) labels[i] = (size_t)(h->base + ids[i]);
    std::vector<double> d(h->len);
    VecSimAdhocBfCtx *actx = VecSimIndex_AdhocBfCtx_New(index, query);
    if (!actx) throw std::runtime_error(last_error());
    VecSimIndex_AdhocBfCtx_GetExactDistances(actx, labels.data(), d.data(), h->len);
    VecSimIndex_AdhocBfCtx_Free(actx);
    std::vector<uint32_t> order;
    for (uint32_t i = 0; i < h->len; i++)
      if (!std::isnan(d[i])) order.push_back(i);  // NaN: the doc has no vector (hybrid_reader.c:317-320)
    const size_t take = std::min<size_t>(k, order.size());
    std::partial_sort(order.begin(), order.begin() + take, order.end(),
                      [&](uint32_t a, uint32_t b) { return d[a] != d[b] ? d[a] < d[b] : a < b; });
    for (size_t i = 0; i < take; i++) {
      if (doc_ids_out) doc_ids_out[i] = labels[order[i]];
      if (dist_out) dist_out[i] = d[order[i]];
    }
    return (long)take;
  }
  if (f->device != h->device) throw std::runtime_error("hits and index live on different devices");
  if (f->key_bytes != 4) throw std::runtime_error("RSGPU_Hits_KnnRerank: FLOAT64 indexes are served by VecSimIndex_AdhocBfCtx_GetExactDistances");
  f->flush_if_needed();
  std::shared_lock<std::shared_mutex> g(f->mu);
  HIP_CHECK(hipSetDevice(h->device));
  if (!h->len || !k) return 0;
  CtxLease c(h->device);
  Scratch &sc = scratch(h->device);
  sc.rows.ensure(h->len);
  sc.dists.ensure(h->len);
  sc.keys32.ensure(h->len);
  f->upload_query(c.c, query, true);
  StageTimer tk(c.c, 4);
  // label -> row on the device (label_table.hpp): identity arithmetic or the direct table.  A multi-value index off identity
  // labelling chains a label's rows: the gather's distance of the first row, then the minimum over the chain.
  LabelRows L;
  const bool on_device = f->device_label_rows(&L);
  const bool chain = on_device && L.next != nullptr;
  if (f->multi && (!on_device || (chain && !knn_chain_supported(f->ktype, f->kmetric, (uint32_t)(f->stride() / 16))))) {
    // multi-value index whose labels have no device table (or a type without a chain kernel): a document's distance is the
    // MINIMUM over its vectors, as GetDistanceFrom / the ad-hoc gather give it (FlatIndex::gather expands every label)
    const std::vector<uint32_t> &ids = h->host_ids();
    std::vector<size_t> labels(ids.size());
    for (size_t i = 0; i < ids.size(); i++) labels[i] = (size_t)(h->base + ids[i]);
    std::vector<double> d(h->len);
    f->gather(c.c, labels.data(), h->len, d.data());
    tk.stop();
    std::vector<uint32_t> order;
    order.reserve(h->len);
    for (uint32_t i = 0; i < h->len; i++)
      if (!std::isnan(d[i])) order.push_back(i);  // NaN: the doc has no vector (hybrid_reader.c:317-320)
    const size_t take = std::min<size_t>(k, order.size());
    std::partial_sort(order.begin(), order.begin() + take, order.end(),
                      [&](uint32_t a, uint32_t b) { return d[a] != d[b] ? d[a] < d[b] : a < b; });
    for (size_t i = 0; i < take; i++) {
      if (doc_ids_out) doc_ids_out[i] = h->base + ids[order[i]];
      if (dist_out) dist_out[i] = d[order[i]];
    }
    return (long)take;
  }
  if (on_device) {
    launch_labels_to_rows(h->ids.p, h->len, h->base, L, sc.rows.p, c->stream);
  } else {  // (no device form of the label map: cannot happen since round 6 -- every mode of label_table.hpp has one)
    const std::vector<uint32_t> &ids = h->host_ids();
    std::vector<uint32_t> rows(h->len);
    for (uint32_t i = 0; i < h->len; i++) rows[i] = f->first_row_of(h->base + ids[i]);
    HIP_CHECK(hipMemcpyAsync(sc.rows.p, rows.data(), h->len * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
  }
  launch_gather(f->device_rows(), f->stride(), (uint32_t)f->dim, f->ktype, f->kmetric, sc.rows.p, h->len, c->d_query,
                sc.dists.p, c->stream);
  if (chain) launch_knn_chain_min(f->device_rows(), f->stride(), f->ktype, f->kmetric, sc.rows.p, h->len, nullptr, L, c->d_query, sc.dists.p, c->stream);
  launch_dist_to_keys(sc.dists.p, h->len, sc.keys32.p, c->stream);
  HIP_CHECK(hipGetLastError());
  std::vector<Hit> hits;
  select_keys32(c.c, sc.keys32.p, h->len, (uint32_t)std::min<size_t>(k, h->len), hits);
  tk.stop();
  const std::vector<uint32_t> &ids = h->host_ids();
  long out = 0;
  for (const Hit &hit : hits) {
    if ((uint32_t)hit.key == 0xFFFFFFFFu) continue;  // NaN: the doc has no vector (hybrid_reader.c:317-320)
    if (doc_ids_out) doc_ids_out[out] = h->base + ids[hit.row];
    if (dist_out) dist_out[out] = (double)key_to_dist((uint32_t)hit.key);
    out++;
  }
  return out;
  S_CATCH(-1)
}


// ---- the whole hybrid query in one call: one stream per branch, two synchronisations --------------------------------
// decode (cached) -> intersect ............................ stream A, sync #1 (the hit count decides every later launch)
//   branch A: score -> 32-bit prefilter -> survivors' 64-bit keys to pinned memory
//   branch B: labels -> rows -> gather -> keys -> top-k      (runs concurrently on its own stream)
// sync #2.  The stage-by-stage entry points above remain (each a stream sync of its own: 5 + 3 inside the selects).
namespace {
struct FusedEvents {
  int device = -1;
  hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void ensure(int dev) {
    if (device == dev) return;
    for (auto &x : e) {
      if (x) (void)hipEventDestroy(x);
      HIP_CHECK(hipEventCreate(&x));
    }
    device = dev;
  }
};
thread_local FusedEvents tls_events;
constexpr uint32_t kFetchCap = 4096;  // survivors of the score prefilter settled on the host
}  // namespace

static void fill_score_params(ScoreParams &P, const RSGPU_Hits *h, const RSGPU_DocTable *t, const RSGPU_ScoreArgs *a,
                              bool *max_norm) {
  memset(&P, 0, sizeof P);
  *max_norm = a->scorer == RSGPU_SCORER_BM25STD_NORM;
  P.scorer = *max_norm ? (int)RSGPU_SCORER_BM25STD : a->scorer;
  tree_score_params(P, h, t);
  P.avg_doc_len = a->avg_doc_len;
  P.root_weight = a->root_weight;
  P.min_score = a->min_score;
  P.inv_tanh = a->tanh_factor ? 1 / (double)a->tanh_factor : 0.0;
  for (int s = 0; s < h->n_lists; s++) {
    int o = h->order[s];
    P.idf[s] = a->idf ? a->idf[o] : 0.0;
    P.bm25_idf[s] = a->bm25_idf ? a->bm25_idf[o] : 0.0;
    P.weight[s] = a->weight ? a->weight[o] : 1.0;
  }
}

thread_local int tls_hybrid_path = 0;  // how the last RSGPU_HybridQuery of this thread ran: 0 staged, 1 two launches

// The bucket directory of a probed list (see RSGPU_Postings::dir): built on the query's stream right behind the decode, once.
// Buckets of ~32 postings on average: shift = log2 of (doc-id range / (entries / 32)).
static void ensure_bucket_dir(RSGPU_Postings *p, QueryCtx *c) {
  if (p->dir_ready.load(std::memory_order_acquire) || !p->n_entries) return;
  std::lock_guard<std::mutex> g(p->decode_mu);
  if (p->dir_ready.load(std::memory_order_relaxed)) return;
  const uint64_t range = p->last - p->base + 1;  // ids[] hold doc id - base
  uint32_t shift = 0;
  while (shift < 31 && (range >> shift) > (uint64_t)p->n_entries / 32 + 1) shift++;
  const uint64_t dn = ((p->last - p->base) >> shift) + 2;
  if (dn > (1ull << 28)) return;  // (cannot happen with the shift above; the wave-wide searches take such a list)
  p->dir.alloc((size_t)dn);
  p->dir_shift = shift;
  p->dir_n = (uint32_t)dn;
  launch_build_bucket_dir(p->ids.p, p->n_entries, shift, p->dir.p, p->dir_n, c->stream);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(c->stream));  // (once per list: other streams read it from now on)
  p->dir_ready.store(true, std::memory_order_release);
}

#define RSGPU_SEARCH_ABI_INTERNAL 1
#include "hybrid_entry.hpp"
#undef RSGPU_SEARCH_ABI_INTERNAL

// reference src/result_processor.c:2549-2571 (window, ranks), src/hybrid/hybrid_scoring.c:41-84
long RSGPU_HybridFuse(int scoring, double rrf_constant, const double *weights, int metric, const uint64_t *search_ids,
                      const double *search_scores, size_t n_search, const uint64_t *vec_ids, const double *vec_scores,
                      size_t n_vec, size_t window, size_t top_n, uint64_t *doc_ids_out, double *scores_out) {
  S_TRY
  if (scoring != RSGPU_HYBRID_RRF && scoring != RSGPU_HYBRID_LINEAR) throw std::runtime_error("RSGPU_HybridFuse: unknown scoring");
  if (scoring == RSGPU_HYBRID_LINEAR && !weights) throw std::runtime_error("RSGPU_HybridFuse: LINEAR needs two weights");
  if (window > kFuseMaxWindow) throw std::runtime_error("RSGPU_HybridFuse: window is limited to 4096 per upstream");
  const uint32_t na = (uint32_t)std::min(n_search, window), nb = (uint32_t)std::min(n_vec, window);
  if ((na && (!search_ids || (scoring == RSGPU_HYBRID_LINEAR && !search_scores))) ||
      (nb && (!vec_ids || (scoring == RSGPU_HYBRID_LINEAR && !vec_scores))))
    throw std::runtime_error("RSGPU_HybridFuse: NULL list");
  if (!na && !nb) return 0;
  std::string why;
  if (!device_available(&why)) throw std::runtime_error(why);
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  CtxLease c(dev);
  Scratch &sc = scratch(dev);
  const uint32_t m = na + nb;
  sc.fuse.ensure((size_t)m * 4 * 8 + 8);  // ids_in | scores_in | ids_out | scores_out | count
  uint64_t *d_ids = sc.fuse.p, *d_oid = d_ids + 2 * (size_t)m;
  double *d_sc = reinterpret_cast<double *>(d_ids + m), *d_osc = reinterpret_cast<double *>(d_ids + 3 * (size_t)m);
  uint32_t *d_cnt = reinterpret_cast<uint32_t *>(d_ids + 4 * (size_t)m);
  std::vector<double> zeros(m, 0.0);
  HIP_CHECK(hipMemcpyAsync(d_ids, search_ids, na * 8, hipMemcpyHostToDevice, c->stream));
  HIP_CHECK(hipMemcpyAsync(d_ids + na, vec_ids, nb * 8, hipMemcpyHostToDevice, c->stream));
  HIP_CHECK(hipMemcpyAsync(d_sc, search_scores ? (const void *)search_scores : (const void *)zeros.data(), na * 8,
                           hipMemcpyHostToDevice, c->stream));
  HIP_CHECK(hipMemcpyAsync(d_sc + na, vec_scores ? (const void *)vec_scores : (const void *)zeros.data(), nb * 8,
                           hipMemcpyHostToDevice, c->stream));
  FuseParams p{scoring, rrf_constant, weights ? weights[0] : 0.0, weights ? weights[1] : 0.0, metric,
               d_ids, d_ids + na, d_sc, d_sc + na, na, nb, d_oid, d_osc, d_cnt};
  launch_hybrid_fuse(p, c->stream);
  HIP_CHECK(hipGetLastError());
  uint32_t cnt = 0;
  HIP_CHECK(hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  const size_t out = std::min<size_t>(cnt, top_n);
  if (out && doc_ids_out) HIP_CHECK(hipMemcpy(doc_ids_out, d_oid, out * 8, hipMemcpyDeviceToHost));
  if (out && scores_out) HIP_CHECK(hipMemcpy(scores_out, d_osc, out * 8, hipMemcpyDeviceToHost));
  return (long)out;
  S_CATCH(-1)
}

// reference src/redisearch_rs/idf/src/lib.rs:67-108
double RSGPU_CalculateIDF(size_t total_docs, size_t term_docs) {
  if (!term_docs) term_docs = 1;
  double value = 1.0 + (double)(total_docs + 1) / (double)term_docs;
  return (double)std::ilogb(value);
}
double RSGPU_CalculateIDF_BM25(size_t total_docs, size_t term_docs) {
  if (total_docs < term_docs) total_docs = term_docs;
  double total = (double)total_docs, term = (double)term_docs;
  return std::log(1.0 + (total - term + 0.5) / (term + 0.5));
}

int RSGPU_HybridQueryPath(void) { return tls_hybrid_path; }
void RSGPU_GetHybridCoalesceStats(uint64_t out[5], int reset) {
  HybCoalesceStats &c = hyb_coalesce_stats();
  if (out) {
    out[0] = c.solo.load();
    out[1] = c.grids.load();
    out[2] = c.grid_queries.load();
    out[3] = c.queued.load();
    out[4] = c.relaunched.load();
  }
  if (reset) c.solo = c.grids = c.grid_queries = c.queued = c.relaunched = 0;
}

// diagnostics (knob hybrid_trace): the phase clock of every tile of the calling thread's last two-launch query, [tiles][9]
// readings of the 100 MHz device clock; returns the number of tiles (0: no trace)
long RSGPU_HybridTrace(uint64_t *out, size_t cap_tiles) {
  S_TRY
  Scratch &sc = tls_scratch;
  const size_t n = std::min<size_t>(sc.hyb_trace_tiles, cap_tiles);
  if (n && out) HIP_CHECK(hipMemcpy(out, sc.hyb_trace.p, n * kHybTracePhases * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return (long)n;
  S_CATCH(-1)
}

void RSGPU_SearchProfile(double *decode_ms, double *intersect_ms, double *score_ms, double *topn_ms, double *knn_ms) {
  if (decode_ms) *decode_ms = prof_ms[0];
  if (intersect_ms) *intersect_ms = prof_ms[1];
  if (score_ms) *score_ms = prof_ms[2];
  if (topn_ms) *topn_ms = prof_ms[3];
  if (knn_ms) *knn_ms = prof_ms[4];
}

}  // extern "C"
