// hybrid_entry.hpp -- the hybrid query entry points of search_abi.cpp (RSGPU_HybridQuery, RSGPU_HybridTreeQuery,
// RSGPU_HybridTreeNodesQuery) and what they share: the plan (validation, locking, label view, deadline, staged tail), the two forms
// of the tile path (hybrid_two_launches, hybrid_general) and the tile-built hit lists of RSGPU_IntersectEx / RSGPU_EvalTree.
// An IMPLEMENTATION INCLUDE: textually part of search_abi.cpp (it uses that file's internal types -- RSGPU_Postings, RSGPU_Hits,
// Scratch, the sources of a combine -- and must not be included anywhere else); split out in round 6 so that neither half is a
// 3 000-line file.  Reference: src/iterators/hybrid_reader.c:309-327,625 (the hybrid iterator and its child filter).
#pragma once
#ifndef RSGPU_SEARCH_ABI_INTERNAL
#error "hybrid_entry.hpp is an implementation include of search_abi.cpp"
#endif

// ---- what the three hybrid entry points share (round 6: one plan instead of three copies of the same prologue) ----------------
// The query's deadline: the reference polls TimedOut_WithCtx per candidate (src/iterators/hybrid_reader.c:311, src/util/timeout.h:
// 57-100) and its iterators return ITERATOR_TIMEOUT; VecSim polls timeoutCallback(queryParams->timeoutCtx).  Here the callback of
// RSGPU_HybridQueryArgs is polled at entry, while the host waits for the device (hyb_wait) and between the stages of the staged
// forms.  A poll that fires first waits for what the query has in flight -- its pinned flags and scratch go back to the pools with
// the leases -- and then unwinds to the entry point, which answers RSGPU_TIMED_OUT with empty outputs.
namespace {
struct QueryTimedOut {
  int unused = 0;
};
struct TileShapeRefused {  // hybrid_general: a query shape the tile kernel's fixed arrays do not hold -> the staged form
  int unused = 0;
};
thread_local const RSGPU_HybridQueryArgs *tls_query = nullptr;
struct QueryScope {
  const RSGPU_HybridQueryArgs *prev;
  explicit QueryScope(const RSGPU_HybridQueryArgs *a) : prev(tls_query) { tls_query = a; }
  ~QueryScope() { tls_query = prev; }
};
inline bool deadline_passed() {
  const RSGPU_HybridQueryArgs *a = tls_query;
  return a && a->timeout_cb && a->timeout_cb(a->timeout_ctx) != 0;
}
inline void poll_deadline(QueryCtx *ca, QueryCtx *cb) {
  if (!deadline_passed()) return;
  if (ca) (void)hipStreamSynchronize(ca->stream);
  if (cb) (void)hipStreamSynchronize(cb->stream);
  throw QueryTimedOut();
}

// validation, the branches wanted, the device, the index behind the KNN branch, outputs zeroed, the first poll of the deadline
struct HybridPlan {
  RSGPU_HybridQueryArgs *a;
  const char *who;
  bool want_score, want_knn;
  int device;
  FlatIndex *f;
  QueryScope scope;
  HybridPlan(const char *who_, RSGPU_HybridQueryArgs *a_, RSGPU_Postings *const *lists, size_t n_lists) : a(a_), who(who_), scope(a_) {
    check_lists(who, lists, n_lists);
    want_score = a->table && a->score && a->top_n;
    want_knn = a->index && a->query && a->k;
    device = lists[0]->device;
    f = want_knn ? a->index->flat : nullptr;
    if (f && f->device != device) throw std::runtime_error(std::string(who) + ": postings and index live on different devices");
    if (want_score && a->table->device != device)
      throw std::runtime_error(std::string(who) + ": postings and document table live on different devices");
    a->n_hits = a->n_top = a->n_knn = 0;
    if (a->hits_out) *a->hits_out = nullptr;
    tls_hybrid_path = 0;
    HIP_CHECK(hipSetDevice(device));
    poll_deadline(nullptr, nullptr);  // (at least once, however small the query: SURVEY.md App. B-7)
    if (f) f->flush_if_needed();
  }
  bool norm() const { return want_score && a->score->scorer == RSGPU_SCORER_BM25STD_NORM; }
  uint32_t top_n_launched() const { return want_score ? (uint32_t)a->top_n + (norm() ? 1u : 0u) : 0u; }
  // the stage-by-stage tail over a hit list: the entry points a caller would use on it (each takes the index's locks itself)
  int staged(std::unique_ptr<RSGPU_Hits> h) {
    tls_hybrid_path = 0;
    if (!h) return -1;
    a->n_hits = h->len;
    poll_deadline(nullptr, nullptr);
    if (want_score) {
      if (RSGPU_Hits_Score(h.get(), a->table, a->score, nullptr) != 0) return -1;
      const long nt = RSGPU_Hits_TopN(h.get(), a->top_n, a->top_ids, a->top_scores);
      if (nt < 0) return -1;
      a->n_top = (size_t)nt;
      poll_deadline(nullptr, nullptr);
    }
    if (want_knn) {
      const long nk = RSGPU_Hits_KnnRerank(h.get(), a->index, a->query, a->k, a->knn_ids, a->knn_dists);
      if (nk < 0) return -1;
      a->n_knn = (size_t)nk;
    }
    if (a->hits_out) *a->hits_out = h.release();
    return 0;
  }
};

// the leases, scratch and profiling events of a tile-path attempt; the index's shared lock and the label view it covers
struct HybridTileRun {
  CtxLease ca, cb;
  Scratch &sc;
  bool prof;
  FusedEvents &ev;
  LabelRows knn_rows{};
  std::shared_lock<std::shared_mutex> index_lock;
  bool labels_ok = true;  // false: a multi-value chain over a type without a chain kernel (the staged KNN expands on the host)
  explicit HybridTileRun(const HybridPlan &p)
      : ca(p.device), cb(p.device), sc(scratch(p.device)), prof(scan_profile().enabled.load(std::memory_order_relaxed) != 0), ev(tls_events) {
    if (prof) ev.ensure(p.device);
    if (p.f) {
      index_lock = std::shared_lock<std::shared_mutex>(p.f->mu);
      labels_ok = p.f->device_label_rows(&knn_rows) &&
                  (!knn_rows.next || knn_chain_supported(p.f->ktype, p.f->kmetric, (uint32_t)(p.f->stride() / 16)));
    }
  }
};
}  // namespace
#define S_CATCH_HYBRID(args)                                               \
  }                                                                        \
  catch (const QueryTimedOut &) {                                          \
    (args)->n_hits = (args)->n_top = (args)->n_knn = 0;                    \
    if ((args)->hits_out && *(args)->hits_out) {                           \
      RSGPU_Hits_Free(*(args)->hits_out);                                  \
      *(args)->hits_out = nullptr;                                         \
    }                                                                      \
    last_error() = "the query's deadline passed";                          \
    return RSGPU_TIMED_OUT;                                                \
  }                                                                        \
  catch (const std::exception &e) {                                        \
    last_error() = e.what();                                               \
    logf(nullptr, "warning", "%s", e.what());                              \
    return -1;                                                             \
  }

// ---- shared by the two forms of the tile path (hybrid_two_launches, hybrid_general) ----
// Scratch for the tiles' lists, the reduce kernel's arguments (answers and completion flags in pinned host memory: ca's for the
// hit count and the scores, cb's for the KNN winners), the flags re-armed.
static void hyb_outputs(Scratch &sc, QueryCtx *ca, QueryCtx *cb, uint32_t n_tiles, uint32_t top_n, uint32_t k, HybridReduceArgs &R) {
  sc.hyb_hits.ensure(n_tiles);
  if (top_n) {
    sc.hyb_skey.ensure((size_t)n_tiles * top_n);
    sc.hyb_sidx.ensure((size_t)n_tiles * top_n);
  }
  if (k) sc.hyb_knn.ensure((size_t)n_tiles * k);
  memset(&R, 0, sizeof R);
  R.n_tiles = n_tiles;
  R.top_n = top_n;
  R.k = k;
  R.surv_cap = (uint32_t)std::min(std::max(scan_tuning().hybrid_surv_cap, 1), 4096);
  R.tile_hits = sc.hyb_hits.p;
  R.part_skey = sc.hyb_skey.p;
  R.part_sidx = sc.hyb_sidx.p;
  R.part_knn = sc.hyb_knn.p;
  ca->ensure_out(std::max<uint32_t>(top_n, 1));
  ca->ensure_gather(std::max<uint32_t>(top_n, 1) + 1);
  cb->ensure_out(std::max<uint32_t>(k, 1));
  cb->ensure_gather(std::max<uint32_t>(k, 1) + 1);
  ca->h_counters[0] = 0;
  ca->h_fcnt[2] = 0;
  cb->h_fcnt[2] = 0;
  R.out_hits = ca->h_counters;
  R.out_skeys = ca->h_out_keys;
  R.out_sids = ca->h_ids;
  R.out_sn = ca->h_fcnt + 2;
  R.out_krows = cb->h_out_rows;
  R.out_kkeys = reinterpret_cast<uint32_t *>(cb->h_out_keys);
  R.out_kids = cb->h_ids;
  R.out_kn = cb->h_fcnt + 2;
  // completion flags the host polls (h_counters[1..3]: pinned, device-visible): hipStreamSynchronize costs several
  // microseconds of a 60 us query once the device is done
  volatile uint32_t *done = ca->h_counters + 1;
  done[0] = top_n ? 0u : 1u;
  done[1] = k ? 0u : 1u;
  done[2] = 0u;
  R.done = ca->h_counters + 1;
}
// the reduce kernel's three flags (may_poll), then -- sync_after: kernels were enqueued behind the reduce kernel -- the stream
static void hyb_wait(QueryCtx *ca, bool may_poll, bool sync_after) {
  volatile uint32_t *done = ca->h_counters + 1;
  bool finished = false;
  if (may_poll && scan_tuning().hybrid_poll) {
    // bounded by TIME (2 ms: forty times a query; a slower one sleeps in the stream sync below), not by an iteration count
    // whose length depends on the host
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; spin++) {
      if (done[0] && done[1] && done[2]) {
        finished = true;
        break;
      }
      cpu_relax();
      if ((spin & 255u) == 255u) poll_deadline(ca, nullptr);  // (waits for the stream before it unwinds: everything rides on ca's)
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!finished || sync_after) HIP_CHECK(hipStreamSynchronize(ca->stream));
  poll_deadline(nullptr, nullptr);
}
// The reduce kernel met more candidates at its bound than it ranks in LDS (an adversarial arrangement of the tiles' lists: mass
// ties across thousands of tiles) and wrote 0xFFFFFFFF instead of a count.  The tiles' lists are still in HBM: the exact radix
// select (select_kernels.hip) takes the k smallest composites of ALL of them -- (score key, position) for the scores: a tile's
// list is sorted by (key, doc id) and the tiles are consecutive doc-id ranges of the driving list, so position order among equal
// keys IS doc-id order; (distance key << 32 | doc id) for the KNN lists -- and fills the pinned answers as the kernel would have.
// Rounds 3-4 re-ran such a query through the ten-kernel staged pipeline, and a query with NOT children (no staged form) failed:
// whether a query succeeded depended on its data (round-4 advisor).
// across_passes (a root union, a root of unions: several passes' tiles side by side): position order is doc-id order inside one
// pass only.  Every entry below the top_n-th key tau is in whatever the order; of the entries AT tau the smallest doc ids are --
// a second select, over their doc ids -- and the list is put in (key, doc id) order on the host.
static void hyb_settle_overflow(Scratch &sc, QueryCtx *ca, QueryCtx *cb, uint32_t n_tiles, uint32_t top_n, uint32_t k,
                                bool across_passes = false) {
  if (!n_tiles) return;
  if (top_n && ca->h_fcnt[2] == 0xFFFFFFFFu) {
    const uint32_t n_hits = ca->h_counters[0];  // (the select below reuses the pinned counters)
    std::vector<Hit> top;
    radix_select(ca, sc.hyb_skey.p, 8, n_tiles * top_n, top_n, Bound(), top, nullptr);
    while (!top.empty() && top.back().key == ~0ull) top.pop_back();  // "none" slots of tiles with fewer hits than top_n
    ca->ensure_gather(top.size() + 1);
    for (size_t i = 0; i < top.size(); i++) ca->h_out_rows[i] = top[i].row;
    if (!top.empty()) {
      launch_gather_u32(sc.hyb_sidx.p, ca->h_out_rows, (uint32_t)top.size(), ca->h_ids, ca->stream);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(ca->stream));
    }
    if (across_passes && !top.empty()) {
      const uint64_t tau = top.back().key;
      std::vector<std::pair<uint64_t, uint32_t>> fin;  // (key, doc id)
      for (size_t i = 0; i < top.size() && top[i].key < tau; i++) fin.emplace_back(top[i].key, ca->h_ids[i]);
      const uint32_t need = (uint32_t)(top.size() - fin.size());
      sc.hyb_tie.ensure((size_t)n_tiles * top_n);
      launch_hybrid_tie_ids(sc.hyb_skey.p, sc.hyb_sidx.p, n_tiles * top_n, tau, sc.hyb_tie.p, ca->stream);
      HIP_CHECK(hipGetLastError());
      std::vector<Hit> ties;
      radix_select(ca, sc.hyb_tie.p, 8, n_tiles * top_n, need, Bound(), ties, nullptr);
      for (const Hit &t : ties)
        if (t.key != ~0ull) fin.emplace_back(tau, (uint32_t)t.key);
      std::sort(fin.begin(), fin.end());
      ca->ensure_gather(fin.size() + 1);
      for (size_t i = 0; i < fin.size(); i++) {
        ca->h_out_keys[i] = fin[i].first;
        ca->h_ids[i] = fin[i].second;
      }
      ca->h_fcnt[2] = (uint32_t)fin.size();
    } else {
      for (size_t i = 0; i < top.size(); i++) ca->h_out_keys[i] = top[i].key;
      ca->h_fcnt[2] = (uint32_t)top.size();
    }
    ca->h_counters[0] = n_hits;
  }
  if (k && cb->h_fcnt[2] == 0xFFFFFFFFu) {
    std::vector<Hit> top;
    radix_select(cb, sc.hyb_knn.p, 8, n_tiles * k, k, Bound(), top, nullptr);
    while (!top.empty() && top.back().key == ~0ull) top.pop_back();
    uint32_t *k32 = reinterpret_cast<uint32_t *>(cb->h_out_keys);
    std::vector<uint64_t> comp(top.size());
    for (size_t i = 0; i < top.size(); i++) comp[i] = top[i].key;  // (h_out_keys is the select's own staging: copy first)
    for (size_t i = 0; i < comp.size(); i++) {
      k32[i] = (uint32_t)(comp[i] >> 32);
      cb->h_ids[i] = (uint32_t)comp[i];
      cb->h_out_rows[i] = (uint32_t)comp[i];
    }
    cb->h_fcnt[2] = (uint32_t)comp.size();
  }
}
// the answers out of pinned memory; false: the reduce kernel met more candidates at its bound than it ranks -- or, BM25STD.NORM,
// the division made a tie across the cut (below): the staged pipeline takes the query.
// norm (SCORER BM25STD.NORM = BM25STD, then every score divided by the largest one: RPMaxScoreNormalizer, src/result_processor.c:
// 1770-1812): the tile kernels ranked BM25STD; the largest score over ALL hits is the first entry's, so the division happens here,
// on the top_n - 1 entries the caller asked for (one more was launched).  x / max is monotone, so the order stands -- except where
// two DIFFERENT scores round to the same quotient: the staged selection ranks the quotients and breaks that tie by doc id.  Inside the
// list that is a re-sort; across the cut (entry top_n - 2 against entry top_n - 1) the list itself might differ: hand the query back.
static bool hyb_collect(RSGPU_HybridQueryArgs *a, uint64_t base, QueryCtx *ca, QueryCtx *cb, uint32_t n_tiles, uint32_t top_n, uint32_t k,
                        bool norm) {
  if (n_tiles && ((top_n && ca->h_fcnt[2] == 0xFFFFFFFFu) || (k && cb->h_fcnt[2] == 0xFFFFFFFFu))) return false;
  a->n_hits = n_tiles ? ca->h_counters[0] : 0;
  if (top_n && n_tiles) {
    const uint32_t got = std::min<uint32_t>(ca->h_fcnt[2], top_n);
    const uint32_t want = norm ? top_n - 1 : top_n;
    std::vector<double> sc(got);
    std::vector<uint32_t> id(ca->h_ids, ca->h_ids + got), ord(got);
    for (uint32_t i = 0; i < got; i++) sc[i] = key2score(ca->h_out_keys[i]);
    std::iota(ord.begin(), ord.end(), 0u);
    if (norm && got) {
      if (std::isnan(sc[0])) return false;
      const double mx = sc[0] > 0.0 ? sc[0] : 0.0;  // max(0, the largest score)
      if (mx != 0.0) {
        std::vector<double> q(got);
        for (uint32_t i = 0; i < got; i++) q[i] = sc[i] / mx;
        if (got > want && q[want - 1] == q[want] && sc[want - 1] != sc[want]) return false;
        std::stable_sort(ord.begin(), ord.begin() + std::min(got, want),
                         [&](uint32_t x, uint32_t y) { return q[x] != q[y] ? q[x] > q[y] : id[x] < id[y]; });
        sc.swap(q);
      }
    }
    const uint32_t n = std::min(got, want);
    for (uint32_t i = 0; i < n; i++) {
      if (a->top_ids) a->top_ids[i] = base + id[ord[i]];
      if (a->top_scores) a->top_scores[i] = sc[ord[i]];
    }
    a->n_top = n;
  }
  if (k && n_tiles) {
    const uint32_t got = std::min<uint32_t>(cb->h_fcnt[2], k);
    const uint32_t *k32 = reinterpret_cast<const uint32_t *>(cb->h_out_keys);
    size_t out = 0;
    for (uint32_t i = 0; i < got; i++) {  // (already in (distance, doc id) order)
      if (k32[i] == 0xFFFFFFFFu) continue;  // NaN: a distance that is not a number ranks nowhere (hybrid_reader.c:317-320)
      if (a->knn_ids) a->knn_ids[out] = base + cb->h_ids[i];
      if (a->knn_dists) a->knn_dists[out] = (double)key_to_dist(k32[i]);
      out++;
    }
    a->n_knn = out;
  }
  return true;
}
static void hyb_profile(bool prof, FusedEvents &ev, uint32_t n_tiles) {
  if (!prof) return;
  float ms = 0;
  prof_ms[0] = prof_ms[2] = prof_ms[4] = 0;
  prof_ms[1] = prof_ms[3] = 0;
  if (n_tiles) {
    if (hipEventElapsedTime(&ms, ev.e[0], ev.e[1]) == hipSuccess) prof_ms[0] = ms;  // decode (nothing when the lists are cached)
    if (hipEventElapsedTime(&ms, ev.e[1], ev.e[2]) == hipSuccess) prof_ms[1] = ms;  // the tile kernel: probe + score + distances
    if (hipEventElapsedTime(&ms, ev.e[2], ev.e[3]) == hipSuccess) prof_ms[3] = ms;  // the reduce kernel (+ the hit list's pack)
  }
}

// ---- the hybrid coalescer (round 6) ----------------------------------------------------------------------------------------------
// RediSearch runs queries from a pool of worker threads (src/util/workers.c:58,104; the hybrid iterator's loop src/iterators/
// hybrid_reader.c:309-327 runs on each of them).  One query's grid is 2 443 tiles on ~1 536 resident workgroups -- a full round and
// a half-empty one -- and its reduce launch waits for the last tile; from eight threads the engine plateaued at 25-26 k QPS (rounds
// 3-5).  Here the two-launch queries of concurrent callers share grids: at most `hybrid_coalesce_depth` grids are in flight per
// device; a caller that arrives below that depth launches at once -- alone, or together with whatever is queued -- and one that
// arrives at it queues its description (HybJob: the tile / reduce arguments over ITS OWN scratch, pinned answers and flags) and
// spins on its own completion flags.  Every grid has an OWNER, the first job of it; when the owner has seen its answers it takes
// the queue (up to kHybBatchMax compatible jobs), launches them as ONE grid + ONE reduce launch on the first one's stream, and
// returns to its caller -- nobody sleeps, nobody is woken.  A single caller never queues: its path is the one of rounds 3-5.
// A job's query vector went up on its own stream; the launching stream waits for the event recorded behind it.  Answers are
// bit-identical to serial ones (fixed slots per tile, total orders: tests/test_gpu_hybrid_coalesce.py).
namespace {
struct HybJob {
  HybridTileLite T;
  HybridReduceArgs R;
  uint32_t n_tiles = 0;
  int sig = -1;  // kernel instantiation the KNN branch needs (type * 8 + metric); -1: no KNN branch -- fits any grid
  size_t lds = 0;
  QueryCtx *ca = nullptr;  // its stream carries the grid this job owns; ev0 = behind the upload of its query
  bool has_event = false;
  bool owner = false;           // written by the launcher before state
  std::atomic<int> state{0};    // 0 queued, 1 launched, -1 the launch failed (the member launches on its own)
};
struct HybCoalescer {
  std::mutex mu;
  std::vector<HybJob *> pending;
  int in_flight = 0;
  std::atomic<int> hint{0};  // in_flight, for the decision to record an event before the lock is taken
};
HybCoalescer *hyb_coalescer_of(int device) {
  static HybCoalescer c[32];
  return &c[device & 31];
}
struct HybCoalesceStats {
  std::atomic<uint64_t> solo{0}, grids{0}, grid_queries{0}, queued{0}, relaunched{0};
};
HybCoalesceStats *hyb_coalesce_stats_ptr() {
  static HybCoalesceStats s;
  return &s;
}
#define hyb_coalesce_stats() (*hyb_coalesce_stats_ptr())
int hyb_depth() { return std::min(std::max(scan_tuning().hybrid_coalesce_depth, 1), 8); }
// compatible jobs off the queue, in arrival order (caller holds the lock)
void hyb_take(HybCoalescer &co, std::vector<HybJob *> &batch, size_t room) {
  int sig = -1;
  for (HybJob *j : batch)
    if (j->sig >= 0) sig = j->sig;
  size_t w = 0;
  for (size_t i = 0; i < co.pending.size(); i++) {
    HybJob *j = co.pending[i];
    if (room && (j->sig < 0 || sig < 0 || j->sig == sig)) {
      if (j->sig >= 0) sig = j->sig;
      batch.push_back(j);
      room--;
    } else {
      co.pending[w++] = j;
    }
  }
  co.pending.resize(w);
}
// ONE grid + ONE reduce launch for the batch, on its first job's stream; every job is told (state) whatever happens
void hyb_launch_batch(std::vector<HybJob *> &batch) {
  HybJob *own = batch.front();
  hipStream_t st = own->ca->stream;
  try {
    std::vector<HybJob *> ord(batch);
    std::stable_sort(ord.begin(), ord.end(), [](const HybJob *x, const HybJob *y) { return x->n_tiles < y->n_tiles; });
    HybridTileBatch B;
    HybridReduceBatch RB;
    memset(&RB, 0, sizeof RB);
    B.n_q = RB.n_q = (uint32_t)ord.size();
    B.interleave = scan_tuning().hybrid_coalesce_interleave ? 1u : 0u;
    size_t lds = 0;
    int sig = -1;
    for (size_t i = 0; i < ord.size(); i++) {
      B.q[i] = ord[i]->T;
      B.n_tiles[i] = ord[i]->n_tiles;
      B.tile_end[i] = 0;
      RB.q[i] = ord[i]->R;
      lds = std::max(lds, ord[i]->lds);
      if (ord[i]->sig >= 0) sig = ord[i]->sig;
      if (ord[i]->has_event && ord[i]->ca->stream != st) HIP_CHECK(hipStreamWaitEvent(st, ord[i]->ca->ev0, 0));
    }
    if (!launch_hybrid_tiles_batch(B, sig >= 0 ? sig / 8 : 0, sig >= 0 ? sig % 8 : 0, lds, st))
      throw std::runtime_error("hybrid coalescer: the shared grid was refused");
    launch_hybrid_reduce_batch(RB, st);
    HIP_CHECK(hipGetLastError());
  } catch (...) {
    for (HybJob *j : batch) {
      j->owner = false;
      j->state.store(-1, std::memory_order_release);
    }
    throw;
  }
  hyb_coalesce_stats().grids++;
  hyb_coalesce_stats().grid_queries += batch.size();
  for (HybJob *j : batch) {
    j->owner = j == own;
    j->state.store(1, std::memory_order_release);
  }
}
// the owner of a grid has its answers: one grid fewer in flight; the queue, if any, goes up as the next one
void hyb_grid_finished(int device) noexcept {
  HybCoalescer &co = *hyb_coalescer_of(device);
  std::vector<HybJob *> next;
  {
    std::lock_guard<std::mutex> g(co.mu);
    co.in_flight--;
    if (!co.pending.empty() && co.in_flight < hyb_depth()) {
      hyb_take(co, next, (size_t)kHybBatchMax);
      co.in_flight++;
    }
    co.hint.store(co.in_flight, std::memory_order_relaxed);
  }
  if (next.empty()) return;
  try {
    hyb_launch_batch(next);
  } catch (const std::exception &e) {  // (the members were told: each launches on its own)
    logf(nullptr, "warning", "%s", e.what());
    std::lock_guard<std::mutex> g(co.mu);
    co.in_flight--;
    co.hint.store(co.in_flight, std::memory_order_relaxed);
  }
}
struct HybGridOwner {  // runs the owner's duty on every way out of the query
  HybJob &job;
  int device;
  ~HybGridOwner() {
    if (job.owner) hyb_grid_finished(device);
  }
};
}  // namespace

// The query in two launches (hybrid_kernels.hip): for callers that do not ask for the hit list.  The caller holds the index
// lock and has checked the shapes (hybrid_tile_supported); ca's stream carries everything, cb lends its pinned buffers to the
// KNN answers; the prepared query is ca->d_query.  false: the reduce kernel met more candidates at its bound than it ranks
// (an adversarial arrangement of the tiles' lists) -- nothing was written, the staged pipeline takes the query.
static bool hybrid_two_launches(RSGPU_HybridQueryArgs *a, FlatIndex *f, const LabelRows &knn_rows, bool want_score, bool want_knn,
                                QueryCtx *ca, QueryCtx *cb, Scratch &sc, bool prof, FusedEvents &ev) {
  const size_t n_lists = a->n_lists;
  std::vector<int> order(n_lists);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a->lists[x]->n_entries < a->lists[y]->n_entries; });
  if (prof) HIP_CHECK(hipEventRecord(ev.e[0], ca->stream));
  for (size_t l = 0; l < n_lists; l++) {
    // (this form reads doc ids and frequencies only: a Full-codec list's masks / offsets index are not decoded for it)
    if (l + 1 < n_lists && decode_pair_on(a->lists[l], a->lists[l + 1], ca, true)) l++;
    else decode_on(a->lists[l], ca, false, true);
  }
  RSGPU_Hits h;  // the tree and the frame of the result (no arrays: nothing is written in hit order)
  h.device = ca->device;
  std::vector<Source> srcs;
  for (size_t s = 0; s < n_lists; s++) srcs.push_back(term_source(a->lists[order[s]], order[s]));
  ListView v;
  const LeafMap m = adopt_sources(&h, srcs, v, 2);
  h.is_union = false;
  const uint32_t n0 = v.len[0];
  const bool norm = want_score && a->score->scorer == RSGPU_SCORER_BM25STD_NORM;  // (one entry more: hyb_collect)
  const uint32_t top_n = want_score ? (uint32_t)a->top_n + (norm ? 1u : 0u) : 0u, k = want_knn ? (uint32_t)a->k : 0u;
  const uint32_t n_tiles = hybrid_tiles(n0);

  HybridTileArgs T;
  memset(&T, 0, sizeof T);
  T.n = v.n;
  for (int l = 0; l < v.n; l++) {
    T.ids[l] = v.ids[l];
    T.freq[l] = m.leaf_freq[l];
    T.len[l] = v.len[l];
    T.add[l] = v.add[l];
  }
  if (scan_tuning().hybrid_dir)
    for (int l = 1; l < v.n; l++) {
      RSGPU_Postings *pl = a->lists[order[l]];
      ensure_bucket_dir(pl, ca);
      if (pl->dir_ready.load(std::memory_order_acquire) && v.ids[l] == pl->ids.p) {
        T.dir[l] = pl->dir.p;
        T.dir_shift[l] = pl->dir_shift;
        T.dir_n[l] = pl->dir_n;
      }
    }
  T.knn_pipeline = (scan_tuning().hybrid_knn_pipeline ? 1 : 0) | (scan_tuning().hybrid_select_split ? 2 : 0);  // (bit 1: tile_select shares the counting among the wavefronts)
  T.top_n = top_n;
  if (want_score) {
    bool max_norm = false;
    fill_score_params(T.P, &h, a->table, a->score, &max_norm);
    T.doc_len = a->table->doc_len.p;
    T.doc_score = a->table->doc_score.p;
    T.max_freq = a->table->max_freq.p;
    T.table_n = a->table->n;
    T.len_score = scan_tuning().hybrid_packed_docs ? reinterpret_cast<const uint2 *>(a->table->len_score.p) : nullptr;
  }
  T.k = k;
  if (want_knn) {
    T.rows = f->device_rows();
    T.stride16 = T.chunks = (uint32_t)(f->stride() / 16);
    T.query = ca->d_query;
    T.ids_base = h.base;
    T.L = knn_rows;
  }
  HybridReduceArgs R;
  hyb_outputs(sc, ca, cb, n_tiles, top_n, k, R);
  sc.hyb_trace_tiles = 0;
  if (scan_tuning().hybrid_trace) {
    sc.hyb_trace.ensure((size_t)(n_tiles + 2) * kHybTracePhases);  // (+ the two branches of the reduce kernel)
    HIP_CHECK(hipMemsetAsync(sc.hyb_trace.p + (size_t)n_tiles * kHybTracePhases, 0, 2 * kHybTracePhases * sizeof(uint64_t), ca->stream));
    T.trace = sc.hyb_trace.p;
    sc.hyb_trace_tiles = n_tiles + 2;
  }
  T.tile_hits = sc.hyb_hits.p;
  T.part_skey = sc.hyb_skey.p;
  T.part_sidx = sc.hyb_sidx.p;
  T.part_knn = sc.hyb_knn.p;
  R.trace = T.trace ? T.trace + (size_t)n_tiles * kHybTracePhases : nullptr;

  if (prof) HIP_CHECK(hipEventRecord(ev.e[1], ca->stream));
  auto launch_alone = [&]() {
    launch_hybrid_tiles(T, f ? f->ktype : 0, f ? f->kmetric : 0, n_tiles, ca->stream);
    if (prof) HIP_CHECK(hipEventRecord(ev.e[2], ca->stream));
    launch_hybrid_reduce(R, ca->stream);
    HIP_CHECK(hipGetLastError());
    if (prof) HIP_CHECK(hipEventRecord(ev.e[3], ca->stream));
  };
  // (the coalescer needs lists any stream may read -- the cached decode publishes them with a stream sync -- and the flags)
  const bool coalesce = n_tiles && scan_tuning().hybrid_coalesce && scan_tuning().cache_decoded && scan_tuning().hybrid_poll && !prof && !T.trace;
  if (n_tiles && !coalesce) {
    launch_alone();
    hyb_wait(ca, !prof && !T.trace, false);
    hyb_settle_overflow(sc, ca, cb, n_tiles, top_n, k);
  } else if (n_tiles) {
    HybCoalescer &co = *hyb_coalescer_of(ca->device);
    HybJob job;
    job.T = hybrid_tile_lite(T, &job.lds);
    job.R = R;
    job.n_tiles = n_tiles;
    job.sig = want_knn ? f->ktype * 8 + f->kmetric : -1;
    job.ca = ca;
    HybGridOwner duty{job, ca->device};
    if (want_knn && co.hint.load(std::memory_order_relaxed) >= hyb_depth()) {  // (probably queued: the upload's event, outside the lock)
      HIP_CHECK(hipEventRecord(ca->ev0, ca->stream));
      job.has_event = true;
    }
    std::vector<HybJob *> batch;
    bool alone = false;
    {
      std::lock_guard<std::mutex> g(co.mu);
      if (co.in_flight < hyb_depth()) {
        co.in_flight++;
        co.hint.store(co.in_flight, std::memory_order_relaxed);
        if (co.pending.empty()) {
          alone = true;
        } else {
          batch.push_back(&job);
          hyb_take(co, batch, (size_t)kHybBatchMax - 1);
        }
      } else {
        if (want_knn && !job.has_event) {
          HIP_CHECK(hipEventRecord(ca->ev0, ca->stream));
          job.has_event = true;
        }
        co.pending.push_back(&job);
        hyb_coalesce_stats().queued++;
      }
    }
    if (alone) {
      job.owner = true;  // (from here on the duty is this thread's, whatever the launch does)
      hyb_coalesce_stats().solo++;
      launch_alone();
      job.state.store(1, std::memory_order_relaxed);
    } else if (!batch.empty()) {
      try {
        hyb_launch_batch(batch);
      } catch (...) {  // (every member was told; the grid's place goes back through this thread's duty)
        job.owner = true;
        throw;
      }
    }
    // the answers: this job's own flags, wherever its grid was launched from
    {
      volatile uint32_t *done = ca->h_counters + 1;
      const auto t0 = std::chrono::steady_clock::now();
      bool timed_out = false, slow = false, left_queue = false;
      for (uint32_t spin = 0;; spin++) {
        if (done[0] && done[1] && done[2]) break;
        const int st = job.state.load(std::memory_order_acquire);
        if (st < 0) {  // the shared launch failed before anything of this job was enqueued: on its own, the classic way
          hyb_coalesce_stats().relaunched++;
          job.state.store(1, std::memory_order_relaxed);
          launch_alone();
        }
        if (slow) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else cpu_relax();
        if ((spin & 255u) == 255u && !timed_out && deadline_passed()) {
          timed_out = true;
          std::lock_guard<std::mutex> g(co.mu);
          auto it = std::find(co.pending.begin(), co.pending.end(), &job);
          if (it != co.pending.end()) {  // still queued: nothing of this job is on the device but its query's upload
            co.pending.erase(it);
            left_queue = true;
            break;
          }
        }
        if ((spin & 1023u) == 1023u && !slow && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) slow = true;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      // (the flags can be up before the launching thread has written owner / state: its last touch of the job is the state)
      while (!left_queue && job.state.load(std::memory_order_acquire) == 0) cpu_relax();
      if (timed_out) {
        (void)hipStreamSynchronize(ca->stream);
        throw QueryTimedOut();
      }
    }
    poll_deadline(nullptr, nullptr);
    hyb_settle_overflow(sc, ca, cb, n_tiles, top_n, k);
  }
  if (!hyb_collect(a, h.base, ca, cb, n_tiles, top_n, k, norm)) return false;
  hyb_profile(prof, ev, n_tiles);
  return true;
}

// ---- the general form of the tile path (hybrid_tree_tile_kernel): a root intersection over terms / unions of terms /
// intersections of terms, max_slop / in_order, slop-dependent scorers over lists with offsets, the hit list itself ----
// One child of the root, in the RESULT's child order (the order RSGPU_EvalTree / intersect_async give the children).
struct HybGroup {
  int op = 0;              // 0 term, 1 union, 2 intersection
  double weight = 1.0;
  std::vector<int> lists;  // the caller's list indices, in the child's own leaf order
  size_t estimate = 0;
  // a child with aggregates of its own (RSGPU_HybridTreeNodesQuery): its result tree over `lists` (post-order, leaf j = lists[j],
  // the child's own node last) and what a hit must hold, as sets of leaves -- any_of: one of them; whole: a nested intersection
  // under a union, absent as a whole unless every term matched; must: the leaves every hit holds (a driver is picked among them)
  bool deep = false;
  bool pred_tree = false;  // (round 6) its shape is beyond the sets below: the kernel folds the match over the result tree
  size_t n_children = 0;
  std::vector<TNode> tree;
  std::vector<uint32_t> any_of, whole, must;
  double key() const {  // (a Not: max_doc_id, last)
    return op == 3 ? 1.0e300 : intersection_sort_key(estimate, op, deep ? n_children : lists.size());
  }
};
// a flat AND: every list a term child, ascending by size, stable (intersection.rs:94-119; intersect_async)
static std::vector<HybGroup> hyb_groups_flat(RSGPU_Postings *const *lists, size_t n_lists, bool in_order = false) {
  std::vector<int> order(n_lists);
  std::iota(order.begin(), order.end(), 0);
  if (!in_order)  // (in_order: the caller's order is the order the terms must appear in)
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return lists[x]->n_entries < lists[y]->n_entries; });
  std::vector<HybGroup> g;
  for (int li : order) {
    HybGroup t;
    t.lists.push_back(li);
    t.estimate = lists[li]->n_entries;
    g.push_back(t);
  }
  return g;
}
// a two-level tree under a root intersection: the children and their leaves in the order RSGPU_EvalTree evaluates them
static std::vector<HybGroup> hyb_groups_tree(const RSGPU_TreeQuery *q, size_t n_lists, const char *who = "RSGPU_HybridTreeQuery",
                                             bool allow_not = true) {
  std::vector<HybGroup> groups;
  for (size_t g = 0; g < q->n_groups; g++) {
    const size_t a = q->group_first[g], b = q->group_first[g + 1];
    if (b <= a || b > n_lists) throw std::runtime_error(std::string(who) + ": bad group_first");
    const int op = q->group_op ? q->group_op[g] : RSGPU_OP_TERM;
    HybGroup t;
    t.weight = q->group_weight ? q->group_weight[g] : 1.0;
    if (op == RSGPU_OP_TERM) {
      if (b - a != 1) throw std::runtime_error(std::string(who) + ": a term group holds exactly one list");
      t.lists.push_back((int)a);
      t.estimate = q->lists[a]->n_entries;
      t.weight = 1.0;  // (a term's own weight stays in RSGPU_ScoreArgs.weight)
    } else if (op == RSGPU_OP_INTERSECT || op == RSGPU_OP_UNION) {
      t.op = op == RSGPU_OP_UNION ? 1 : 2;
      for (size_t l = a; l < b; l++) t.lists.push_back((int)l);
      if (op == RSGPU_OP_INTERSECT) {
        std::stable_sort(t.lists.begin(), t.lists.end(), [&](int x, int y) { return q->lists[x]->n_entries < q->lists[y]->n_entries; });
        t.estimate = q->lists[t.lists[0]]->n_entries;  // num_estimated of an intersection: its smallest child
      } else {
        for (int li : t.lists) t.estimate += q->lists[li]->n_entries;  // ... of a union: the sum
      }
    } else if (op == RSGPU_OP_NOT && allow_not && q->root_op == RSGPU_OP_INTERSECT) {
      t.op = 3;  // excluded lists: no leaf of the result tree, a virtual child of frequency 0
      for (size_t l = a; l < b; l++) t.lists.push_back((int)l);
      t.estimate = ~(size_t)0;  // (a Not's estimate is max_doc_id, never below a real child's: it sorts behind them)
    } else {
      throw std::runtime_error(std::string(who) + ": bad group_op");
    }
    groups.push_back(t);
  }
  // (an intersection iterates its children by ascending estimate; a union keeps the query's order: union_flat.rs)
  if (!q->in_order && q->root_op == RSGPU_OP_INTERSECT)
    std::stable_sort(groups.begin(), groups.end(), [](const HybGroup &x, const HybGroup &y) { return x.key() < y.key(); });
  return groups;
}
// the leaf that drives the probe: the shortest list every hit must hold (a term child, a term of a child intersection);
// -1: every child is a union -- the staged pipeline takes the query
static int hyb_driver(const std::vector<HybGroup> &groups, RSGPU_Postings *const *lists, uint32_t *n0_out) {
  int best = -1;
  uint32_t n0 = 0;
  for (const HybGroup &g : groups) {
    if (g.deep) {
      for (uint32_t m : g.must)
        for (size_t j = 0; j < g.lists.size(); j++)
          if (((m >> j) & 1u) && (best < 0 || lists[g.lists[j]]->n_entries < n0)) {
            best = g.lists[j];
            n0 = lists[best]->n_entries;
          }
      continue;
    }
    if (g.op == 0 || g.op == 2)
      for (int li : g.lists)
        if (best < 0 || lists[li]->n_entries < n0) {
          best = li;
          n0 = lists[li]->n_entries;
        }
  }
  *n0_out = n0;
  return best;
}

// No term every hit holds -- every child of the root intersection is a union, `(run|running|ran) (shoe|shoes)`, the stemmer's
// expansions (round 5): the child union with the fewest postings drives, one pass of the tile kernel per term of it; a document an
// EARLIER term of that union holds belongs to that term's pass.  -1: no child is a plain union of terms.
static int hyb_union_driver_group(const std::vector<HybGroup> &groups, RSGPU_Postings *const *lists, uint32_t *tiles_out) {
  int best = -1;
  uint64_t best_n = 0, best_tiles = 0;
  for (size_t g = 0; g < groups.size(); g++) {
    if (groups[g].op != 1 || groups[g].deep) continue;
    uint64_t n = 0, tiles = 0;
    for (int li : groups[g].lists) {
      n += lists[li]->n_entries;
      tiles += hybrid_tiles(lists[li]->n_entries);
    }
    if (best < 0 || n < best_n) {
      best = (int)g;
      best_n = n;
      best_tiles = tiles;
    }
  }
  if (tiles_out) *tiles_out = (uint32_t)std::min<uint64_t>(best_tiles, 0xFFFFFFFFull);
  return best;
}
// the tiles of a root intersection: its driver's, or -- no term every hit holds, nobody wants the hit list (its order would
// interleave the passes) -- those of the union that drives it; 0: no form on the tile kernel
static uint32_t hyb_intersection_tiles(const std::vector<HybGroup> &groups, RSGPU_Postings *const *lists, bool hits_wanted) {
  uint32_t n0 = 0, tiles = 0;
  if (hyb_driver(groups, lists, &n0) >= 0) return hybrid_tiles(n0);
  (void)hits_wanted;  // (round 6: the passes' runs are merged by doc id -- hybrid_hits_merge_kernel)
  if (hyb_union_driver_group(groups, lists, &tiles) < 0) return 0;
  return tiles;
}

// the shortest list of one child (a term, or the terms of a child intersection); -1: a union child has no list every hit holds
static int hyb_group_driver(const HybGroup &g, RSGPU_Postings *const *lists, uint32_t *n0_out) {
  int best = -1;
  uint32_t n0 = 0;
  if (g.op == 0 || g.op == 2)
    for (int li : g.lists)
      if (best < 0 || lists[li]->n_entries < n0) {
        best = li;
        n0 = lists[li]->n_entries;
      }
  *n0_out = n0;
  return best;
}
// tiles of a root UNION (one pass per child, the child's shortest list drives); 0: a child is a union / empty -- no such form
static uint32_t hyb_union_tiles(const std::vector<HybGroup> &groups, RSGPU_Postings *const *lists) {
  uint64_t total = 0;
  for (const HybGroup &g : groups) {
    uint32_t n0 = 0;
    if (hyb_group_driver(g, lists, &n0) < 0 || !n0) return 0;
    total += hybrid_tiles(n0);
  }
  return (uint32_t)std::min<uint64_t>(total, 0xFFFFFFFFull);
}

// The caller holds the index lock and has checked the shapes (hybrid_tree_supported over the driver's tiles, every list
// non-empty, <= kHybTreeMaxLists lists, not BM25STD.NORM).  hits_out (may be NULL): receives the hit list.  false: BM25STD.NORM
// met a tie across its cut (hyb_collect) -- nothing was handed out, the staged pipeline takes the query.
// root_union (round 5; `a | b`, `(a b) | (c d)`, `a | (b c)`: union_flat.rs:223-320): the children in the QUERY's order, one
// pass of the tile kernel per child -- the child's shortest list drives, every other list is probed; a document an EARLIER
// child matches belongs to that child's pass (veto_all), a LATER child intersection counts only when it matches as a whole
// (opt_all) -- all passes write their tiles' fixed slots side by side and ONE reduce kernel ranks them: the composites are
// total orders and every hit is reported by exactly one pass.  No hit list (its order would interleave the passes), no
// slop-dependent scorer (a union result's slop depends on which children matched): the caller checks.
static bool hybrid_general(RSGPU_HybridQueryArgs *a, RSGPU_Postings *const *lists, const std::vector<HybGroup> &groups, long max_slop,
                           int in_order, RSGPU_Hits **hits_out, FlatIndex *f, const LabelRows &knn_rows, bool want_score, bool want_knn,
                           QueryCtx *ca, QueryCtx *cb, Scratch &sc, bool prof, FusedEvents &ev, bool root_union = false) {
  if (prof) HIP_CHECK(hipEventRecord(ev.e[0], ca->stream));
  // (doc ids + frequencies only -- RSGPU_Postings::decoded_lean -- unless the query walks the term offsets: a window, a scorer
  // that divides by the slop; a caller that takes the hit list may ask for term records later: whole)
  bool lean = !hits_out && max_slop < 0 && !in_order && !(want_score && slop_dependent(a->score->scorer == RSGPU_SCORER_BM25STD_NORM ? (int)RSGPU_SCORER_BM25STD : a->score->scorer));
  for (const HybGroup &g : groups)
    for (int li : g.lists) decode_on(lists[li], ca, false, lean);
  // the result's tree, frame and leaf columns: the children as sources (an aggregate child only lends its shape here -- its
  // lists are probed one by one, no hit list of its own is ever built)
  std::unique_ptr<RSGPU_Hits> hp(new RSGPU_Hits());
  RSGPU_Hits &h = *hp;
  h.device = ca->device;
  std::vector<Source> srcs;
  for (const HybGroup &g : groups) {
    if (g.op == 0) {
      srcs.push_back(term_source(lists[g.lists[0]], g.lists[0]));
      continue;
    }
    if (g.op == 3) {  // a NOT child: a virtual result -- no leaves, contributes weight * 0 -- that still counts as a child
      Source s;
      s.op = 2;
      s.weight = g.weight;
      s.n_leaves = 0;
      s.len = 0;
      s.tree.push_back(TNode{2, 0, 0, g.weight});
      srcs.push_back(s);
      continue;
    }
    Source s;
    s.op = g.op;
    s.weight = g.weight;
    s.n_leaves = (int)g.lists.size();
    s.len = (uint32_t)std::min<size_t>(std::max<size_t>(g.estimate, 1), 0xFFFFFFF0ull);
    s.first = ~0ull;
    for (size_t j = 0; j < g.lists.size(); j++) {
      RSGPU_Postings *pl = lists[g.lists[j]];
      s.freq[j] = pl->cd.freq >= 0 ? pl->freqs.p : nullptr;
      s.src[j] = pl;
      s.orig[j] = g.lists[j];
      s.first = std::min(s.first, pl->first_id);
      s.last = std::max(s.last, pl->last);
      if (!g.deep) s.tree.push_back(TNode{0, (uint8_t)j, 0, 1.0});
    }
    s.base = s.first;  // (only v.add of the ListView uses it: not read here)
    if (g.deep) s.tree = g.tree;
    else s.tree.push_back(TNode{(uint8_t)g.op, 0, (uint16_t)g.lists.size(), g.weight});
    srcs.push_back(s);
  }
  ListView v;
  const LeafMap m = adopt_sources(&h, srcs, v, root_union ? 1 : 2);
  h.is_union = root_union;
  const int n = h.n_lists;  // leaves
  const bool norm = want_score && a->score->scorer == RSGPU_SCORER_BM25STD_NORM;  // (one entry more: hyb_collect)
  const uint32_t top_n = want_score ? (uint32_t)a->top_n + (norm ? 1u : 0u) : 0u, k = want_knn ? (uint32_t)a->k : 0u;

  // the passes: one (root intersection: the shortest required list drives), or one per child of a root union
  struct Pass {
    int driver;  // a caller's list index
    int group;   // root union: the child this pass belongs to
    uint32_t n0, tiles, first_tile;
  };
  std::vector<Pass> passes;
  uint32_t n_tiles = 0;
  int union_driven = -1;  // the child union whose terms drive, one pass each (no term every hit holds)
  if (root_union) {
    for (size_t g = 0; g < groups.size(); g++) {
      Pass p{-1, (int)g, 0, 0, n_tiles};
      p.driver = hyb_group_driver(groups[g], lists, &p.n0);
      if (p.driver < 0 || !p.n0) throw std::runtime_error("hybrid query: a root union's children must be terms or intersections of terms");
      p.tiles = hybrid_tiles(p.n0);
      n_tiles += p.tiles;
      passes.push_back(p);
    }
  } else {
    Pass p{-1, -1, 0, 0, 0};
    p.driver = hyb_driver(groups, lists, &p.n0);
    if (p.driver >= 0) {
      p.tiles = n_tiles = hybrid_tiles(p.n0);
      passes.push_back(p);
    } else {  // every child is a union: one of them drives, term by term (hyb_union_driver_group)
      union_driven = hyb_union_driver_group(groups, lists, nullptr);
      if (union_driven < 0) throw std::runtime_error("hybrid query: no term or union of terms to drive the tile kernel");
      for (int li : groups[union_driven].lists) {
        Pass q{li, union_driven, lists[li]->n_entries, hybrid_tiles(lists[li]->n_entries), n_tiles};
        n_tiles += q.tiles;
        passes.push_back(q);
      }
    }
  }

  std::vector<RSGPU_Postings *> excluded;  // NOT children's lists: probed behind the leaves, no column of their own
  for (const HybGroup &g : groups)
    if (g.op == 3)
      for (int li : g.lists)
        if (lists[li]->n_entries) excluded.push_back(lists[li]);  // (an empty list excludes nothing)
  if (n + (int)excluded.size() > kHybTreeMaxLists) throw std::runtime_error("hybrid query: more than eight lists");
  if (root_union && !excluded.empty()) throw std::runtime_error("hybrid query: a root union on the tile path has no NOT children");
  HybridTreeArgs T;
  memset(&T, 0, sizeof T);
  T.n = n + (int)excluded.size();
  T.n_leaves = n;
  for (int t = 0; t < n; t++) {
    T.lfreq[t] = m.leaf_freq[t];
    const RSGPU_Postings *pl = h.src[t];
    T.O.bytes[t] = pl->bytes.p;
    T.O.off_pos[t] = pl->has_offsets() ? pl->off_pos.p : nullptr;
    T.O.off_len[t] = pl->has_offsets() ? pl->off_len.p : nullptr;
  }
  T.X = tree_prox(&h, max_slop, in_order);
  // (combine_and: the filter runs when a window is asked for, the root has more than one child and some list stores offsets)
  T.prox_filter = (!root_union && (max_slop >= 0 || in_order) && h.n_groups > 1 && h.with_offsets) ? 1 : 0;
  T.knn_pipeline = (scan_tuning().hybrid_knn_pipeline ? 1 : 0) | (scan_tuning().hybrid_select_split ? 2 : 0);  // (bit 1: tile_select shares the counting among the wavefronts)
  T.top_n = top_n;
  if (want_score) {
    bool max_norm = false;
    fill_score_params(T.P, &h, a->table, a->score, &max_norm);
    // (hit_slops: the per-hit slop exists when some list stores offsets and the root has two children or more)
    T.prox_slop = (!root_union && slop_dependent(T.P.scorer) && h.with_offsets && h.n_groups >= 2) ? 1 : 0;
    T.doc_len = a->table->doc_len.p;
    T.doc_score = a->table->doc_score.p;
    T.max_freq = a->table->max_freq.p;
    T.table_n = a->table->n;
    T.len_score = scan_tuning().hybrid_packed_docs ? reinterpret_cast<const uint2 *>(a->table->len_score.p) : nullptr;
  }
  T.k = k;
  if (want_knn) {
    T.rows = f->device_rows();
    T.stride16 = T.chunks = (uint32_t)(f->stride() / 16);
    T.query = ca->d_query;
    T.ids_base = h.base;
    T.L = knn_rows;
  }
  for (const HybGroup &g : groups) T.tree_pred |= g.pred_tree ? 1 : 0;
  if (T.tree_pred) {
    if (!want_score) tree_score_params(T.P, &h, nullptr);  // (the tree alone: the match is folded over it)
    if (T.P.n_nodes <= 0 || root_union) return false;       // (no node array: the staged form)
  }
  HybridReduceArgs R;
  hyb_outputs(sc, ca, cb, n_tiles, top_n, k, R);
  sc.hyb_trace_tiles = 0;
  const uint32_t stride = n_tiles * 1024u;
  if (hits_out) {
    sc.hyb_hit_ids.ensure(stride);
    sc.hyb_hit_freqs.ensure((size_t)stride * n);
    if (h.with_offsets) sc.hyb_hit_epos.ensure((size_t)stride * n);
    T.hit_ids = sc.hyb_hit_ids.p;
    T.hit_freqs = sc.hyb_hit_freqs.p;
    T.hit_epos = h.with_offsets ? sc.hyb_hit_epos.p : nullptr;
    T.hit_stride = stride;
    // (several passes -- round 6: every pass reports its hits in doc-id order, a doc id once over all passes; the packed runs are
    // merged into one ascending list behind the reduce kernel, hybrid_hits_merge_kernel)
    uint64_t cap = 0;
    for (const Pass &ps : passes) cap += ps.n0;
    h.cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cap, 1), stride);
    if (passes.size() > 1) {
      if (passes.size() > 8) throw std::runtime_error("hybrid query: more than eight passes");  // (<= 8 lists: cannot happen)
      sc.hyb_run_ids.ensure(h.cap);
      sc.hyb_run_freqs.ensure((size_t)h.cap * n);
      if (h.with_offsets) sc.hyb_run_epos.ensure((size_t)h.cap * n);
      sc.hyb_run_start.ensure(16);
    }
    h.ids.alloc(h.cap);
    h.freqs.alloc((size_t)h.cap * n);
    if (h.with_offsets) h.epos.alloc((size_t)h.cap * n);
  }
  ca->h_fcnt[0] = 0;

  // one pass: the lists in probe order (the driver, then the other leaves in leaf order, then the excluded lists), what a hit
  // must hold, where the pass's tiles write
  auto fill_pass = [&](HybridTreeArgs &P, const Pass &ps) {
    P = T;
    int leaf_of_list[kHybTreeMaxLists], list_of_leaf[kHybTreeMaxLists];
    {
      int driver_leaf = -1;
      for (int t = 0; t < n; t++)
        if (driver_leaf < 0 && h.order[t] == ps.driver) driver_leaf = t;
      int l = 1;
      for (int t = 0; t < n; t++) {
        const int slot = t == driver_leaf ? 0 : l++;
        leaf_of_list[slot] = t;
        list_of_leaf[t] = slot;
      }
    }
    for (int l = 0; l < n; l++) {
      const int t = leaf_of_list[l];
      RSGPU_Postings *pl = const_cast<RSGPU_Postings *>(h.src[t]);
      P.ids[l] = pl->ids.p;
      P.len[l] = pl->n_entries;
      P.add[l] = (long long)(pl->base - h.base);  // (two's complement: negative when the list's base lies below the frame's)
      P.leaf_of[l] = (uint8_t)t;
      if (l && scan_tuning().hybrid_dir) {
        ensure_bucket_dir(pl, ca);
        if (pl->dir_ready.load(std::memory_order_acquire)) {
          P.dir[l] = pl->dir.p;
          P.dir_shift[l] = pl->dir_shift;
          P.dir_n[l] = pl->dir_n;
        }
      }
    }
    for (size_t x = 0; x < excluded.size(); x++) {
      const int l = n + (int)x;
      RSGPU_Postings *pl = excluded[x];
      P.ids[l] = pl->ids.p;
      P.len[l] = pl->n_entries;
      P.add[l] = (long long)(pl->base - h.base);
      P.leaf_of[l] = 0xFF;
      P.veto |= 1u << l;
      if (scan_tuning().hybrid_dir) {
        ensure_bucket_dir(pl, ca);
        if (pl->dir_ready.load(std::memory_order_acquire)) {
          P.dir[l] = pl->dir.p;
          P.dir_shift[l] = pl->dir_shift;
          P.dir_n[l] = pl->dir_n;
        }
      }
    }
    for (int g = 0; g < h.n_groups; g++) {
      uint32_t all = 0;
      for (int t = h.group_first[g]; t < h.group_first[g + 1]; t++) all |= 1u << list_of_leaf[t];
      if (g == union_driven) {
        // the driving union: this pass's term holds the document (it drives); one that an earlier term holds is that pass's hit
        for (int t = h.group_first[g]; t < h.group_first[g + 1] && h.order[t] != ps.driver; t++) P.veto |= 1u << list_of_leaf[t];
      } else if (T.tree_pred) {
        // (the kernel folds the match over the result tree: no sets)
      } else if (!root_union && groups[g].deep) {
        // a child with aggregates of its own: its sets of leaves, moved to this pass's list slots
        auto slots = [&](uint32_t leaves) {
          uint32_t m = 0;
          for (int t = h.group_first[g]; t < h.group_first[g + 1]; t++)
            if ((leaves >> (t - h.group_first[g])) & 1u) m |= 1u << list_of_leaf[t];
          return m;
        };
        for (uint32_t m : groups[g].any_of) {
          if (P.n_req >= kHybTreeMaxLists) throw TileShapeRefused();  // more than eight required sets: the staged form takes it
          P.req[P.n_req++] = slots(m);
        }
        for (uint32_t m : groups[g].whole) {
          if (P.n_opt_all >= kHybTreeMaxLists) throw TileShapeRefused();  // more than eight nested intersections
          P.opt_all[P.n_opt_all++] = slots(m);
        }
      } else if (!root_union) {
        // what a hit must hold: a term; every term of a child intersection; any term of a child union
        if (h.group_op[g] == 1) {
          P.req[P.n_req++] = all;  // (a NOT child's virtual group has no leaves: nothing required)
        } else {
          for (int t = h.group_first[g]; t < h.group_first[g + 1]; t++) P.req[P.n_req++] = 1u << list_of_leaf[t];
        }
      } else if (g == ps.group) {
        for (int t = h.group_first[g]; t < h.group_first[g + 1]; t++) P.req[P.n_req++] = 1u << list_of_leaf[t];
      } else {
        if (g < ps.group) P.veto_all[P.n_veto_all++] = all;
        // (any OTHER child intersection that does not match as a whole is not in the result -- an earlier one that does is
        // vetoed above)
        if (h.group_op[g] == 2 && h.group_first[g + 1] - h.group_first[g] > 1) P.opt_all[P.n_opt_all++] = all;
      }
    }
    if (P.hit_ids) {  // (the tiles of a pass index their records by their own block numbers)
      P.hit_ids += (size_t)ps.first_tile * 1024u;
      P.hit_freqs += (size_t)ps.first_tile * 1024u;
      if (P.hit_epos) P.hit_epos += (size_t)ps.first_tile * 1024u;
    }
    P.tile_hits = sc.hyb_hits.p + ps.first_tile;
    P.part_skey = sc.hyb_skey.p + (size_t)ps.first_tile * top_n;
    P.part_sidx = sc.hyb_sidx.p + (size_t)ps.first_tile * top_n;
    P.part_knn = sc.hyb_knn.p + (size_t)ps.first_tile * k;
  };

  if (prof) HIP_CHECK(hipEventRecord(ev.e[1], ca->stream));
  if (n_tiles) {
    // every pass's arguments BEFORE the first launch: a shape the kernel's arrays do not hold is handed back to the staged form
    // with nothing in flight (round-5 advisor: the refusal used to surface as an error after earlier passes had been launched)
    std::vector<HybridTreeArgs> filled(passes.size());
    try {
      for (size_t pi = 0; pi < passes.size(); pi++) fill_pass(filled[pi], passes[pi]);
    } catch (const TileShapeRefused &) {
      return false;
    }
    for (size_t pi = 0; pi < passes.size(); pi++)
      launch_hybrid_tree_tiles(filled[pi], f ? f->ktype : 0, f ? f->kmetric : 0, passes[pi].tiles, ca->stream);
    if (prof) HIP_CHECK(hipEventRecord(ev.e[2], ca->stream));
    launch_hybrid_reduce(R, ca->stream);
    if (hits_out && passes.size() == 1) {
      launch_hybrid_hits_pack(sc.hyb_hits.p, n_tiles, n, T.hit_ids, T.hit_freqs, T.hit_epos, stride, h.ids.p, h.freqs.p,
                              h.with_offsets ? h.epos.p : nullptr, h.cap, ca->h_fcnt, ca->stream);
    } else if (hits_out) {
      HybridRuns runs;
      memset(&runs, 0, sizeof runs);
      runs.n = (uint32_t)passes.size();
      for (size_t pi = 0; pi < passes.size(); pi++) runs.first_tile[pi] = passes[pi].first_tile;
      runs.first_tile[passes.size()] = n_tiles;
      runs.run_start = sc.hyb_run_start.p;
      launch_hybrid_hits_pack(sc.hyb_hits.p, n_tiles, n, T.hit_ids, T.hit_freqs, T.hit_epos, stride, sc.hyb_run_ids.p, sc.hyb_run_freqs.p,
                              h.with_offsets ? sc.hyb_run_epos.p : nullptr, h.cap, ca->h_fcnt, ca->stream, &runs);
      launch_hybrid_hits_merge(runs, n, sc.hyb_run_ids.p, sc.hyb_run_freqs.p, h.with_offsets ? sc.hyb_run_epos.p : nullptr, h.cap, h.ids.p,
                               h.freqs.p, h.with_offsets ? h.epos.p : nullptr, h.cap, h.cap, ca->stream);
    }
    HIP_CHECK(hipGetLastError());
    if (prof) HIP_CHECK(hipEventRecord(ev.e[3], ca->stream));
    hyb_wait(ca, !prof, hits_out != nullptr);
    hyb_settle_overflow(sc, ca, cb, n_tiles, top_n, k, root_union || passes.size() > 1);
  }
  if (!hyb_collect(a, h.base, ca, cb, n_tiles, top_n, k, norm)) return false;
  if (hits_out) {
    h.len = n_tiles ? ca->h_fcnt[0] : 0;
    if (h.len != a->n_hits) throw std::runtime_error("RSGPU_HybridQuery: the packed hit list and the hit count disagree");
    *hits_out = hp.release();
  }
  hyb_profile(prof, ev, n_tiles);
  return true;
}

static RSGPU_Hits *eval_tree_tiles(const RSGPU_TreeQuery *q, size_t n_lists) {
  tls_hybrid_path = 0;
  if (!scan_tuning().hybrid_tiles || !scan_tuning().hybrid_tree_tiles || q->root_op != RSGPU_OP_INTERSECT ||
      n_lists > (size_t)kHybTreeMaxLists || !q->group_op)
    return nullptr;
  if (scan_profile().enabled.load(std::memory_order_relaxed)) return nullptr;  // (per-STAGE device times are the staged form's)
  bool aggregate = false;
  for (size_t g = 0; g < q->n_groups; g++) aggregate |= q->group_op[g] != RSGPU_OP_TERM;
  if (!aggregate) return nullptr;  // (a flat AND: the staged intersection is three launches as well)
  for (size_t l = 0; l < n_lists; l++)
    if (!q->lists[l]->n_entries) return nullptr;
  const std::vector<HybGroup> groups = hyb_groups_tree(q, n_lists, "RSGPU_EvalTree", false);
  uint32_t n0 = 0;
  if (hyb_driver(groups, q->lists, &n0) < 0 || !hybrid_tree_supported(0, 0, 1u, hybrid_tiles(n0), 0u, 0u, (int)n_lists)) return nullptr;
  const int device = q->lists[0]->device;
  CtxLease ca(device), cb(device);
  RSGPU_HybridQueryArgs none;
  memset(&none, 0, sizeof none);
  RSGPU_Hits *out = nullptr;
  if (!hybrid_general(&none, q->lists, groups, q->max_slop, q->in_order, &out, nullptr, LabelRows{}, false, false, ca.c, cb.c, scratch(device), false,
                      tls_events))
    return nullptr;
  tls_hybrid_path = 2;
  return out;
}

static RSGPU_Hits *intersect_tiles(RSGPU_Postings *const *lists, size_t n_lists, long max_slop, int in_order) {
  tls_hybrid_path = 0;
  if (!(max_slop >= 0 || in_order) || n_lists < 2 || n_lists > (size_t)kHybTreeMaxLists) return nullptr;
  if (!scan_tuning().hybrid_tiles || !scan_tuning().hybrid_tree_tiles || scan_profile().enabled.load(std::memory_order_relaxed)) return nullptr;
  bool offsets = false;
  uint32_t n0 = 0xFFFFFFFFu;
  for (size_t l = 0; l < n_lists; l++) {
    if (!lists[l]->n_entries) return nullptr;
    offsets |= lists[l]->has_offsets();
    n0 = std::min<uint32_t>(n0, lists[l]->n_entries);
  }
  if (!offsets || !hybrid_tree_supported(0, 0, 1u, hybrid_tiles(n0), 0u, 0u, (int)n_lists)) return nullptr;
  const int device = lists[0]->device;
  CtxLease ca(device), cb(device);
  RSGPU_HybridQueryArgs none;
  memset(&none, 0, sizeof none);
  RSGPU_Hits *out = nullptr;
  if (!hybrid_general(&none, lists, hyb_groups_flat(lists, n_lists, in_order != 0), max_slop, in_order, &out, nullptr, LabelRows{}, false, false, ca.c,
                      cb.c, scratch(device), false, tls_events))
    return nullptr;
  tls_hybrid_path = 2;
  return out;
}

extern "C" int RSGPU_HybridQuery(RSGPU_HybridQueryArgs *a) {
  if (!a || !a->lists || !a->n_lists || a->n_lists > (size_t)kMaxLists) {
    last_error() = "RSGPU_HybridQuery: 1..32 lists";
    return -1;
  }
  S_TRY
  HybridPlan plan("RSGPU_HybridQuery", a, a->lists, a->n_lists);
  const bool want_score = plan.want_score, want_knn = plan.want_knn;
  const int device = plan.device;
  FlatIndex *f = plan.f;
  if (f && f->key_bytes != 4) throw std::runtime_error("RSGPU_HybridQuery: FLOAT64 indexes are not served by the fused path");
  HybridTileRun run(plan);
  CtxLease &ca = run.ca, &cb = run.cb;
  Scratch &sc = run.sc;
  const bool prof = run.prof;
  FusedEvents &ev = run.ev;
  std::unique_ptr<RSGPU_Hits> h(new RSGPU_Hits());
  h->device = device;
  h->n_lists = (int)a->n_lists;

  // Two launches instead of ten (hybrid_kernels.hip) when nobody asked for the hit list and the query has the plain shape:
  // a flat AND of a few term lists, a scorer that needs neither the term offsets nor the maximum over all hits, small N / k.
  // The general form of the tile kernel (round 4) takes what that leaves -- the hit list wanted (a third launch packs it),
  // five to eight lists, scorers that divide by the slop over lists with offsets.  BM25STD.NORM: ranked as BM25STD, divided
  // by the first entry's score on the host (hyb_collect).
  const bool tile_knob = scan_tuning().hybrid_tiles && (want_score || want_knn);
  const bool norm = want_score && a->score->scorer == RSGPU_SCORER_BM25STD_NORM;
  bool slop_offsets = false;
  if (want_score && slop_dependent(a->score->scorer))
    for (size_t l = 0; l < a->n_lists; l++) slop_offsets |= a->lists[l]->has_offsets();
  bool tiles = tile_knob && !a->hits_out && a->n_lists <= (size_t)kHybMaxLists && !slop_offsets && !scan_tuning().hybrid_force_general;
  bool general = tile_knob && !tiles && scan_tuning().hybrid_tree_tiles && a->n_lists <= (size_t)kHybTreeMaxLists;
  uint32_t n0_min = 0xFFFFFFFFu;
  for (size_t l = 0; l < a->n_lists; l++) n0_min = std::min<uint32_t>(n0_min, a->lists[l]->n_entries);
  if (tiles || general) {
    const bool ok = n0_min > 0 && hybrid_tile_supported(f ? f->ktype : 0, f ? f->kmetric : 0, f ? (uint32_t)(f->stride() / 16) : 1u,
                                                        hybrid_tiles(n0_min), want_score ? (uint32_t)a->top_n + (norm ? 1u : 0u) : 0u,
                                                        want_knn ? (uint32_t)a->k : 0u);
    tiles = tiles && ok;
    general = general && ok;
  }

  // the KNN branch's query goes up first, on its own stream: it does not depend on the hits
  // doc id -> row: identity arithmetic, the direct table or -- labels far apart -- the hash table in HBM (label_table.hpp; every
  // form survives deletes, re-adds under new ids, documents without a vector and multi-value labels; round 6: no host translation).
  // The plan's run holds the index's shared lock and the label view it covers.
  const LabelRows &knn_rows = run.knn_rows;
  std::shared_lock<std::shared_mutex> &index_lock = run.index_lock;
  const bool knn_identity = f && run.labels_ok;  // (historic name: the KNN branch translates on the device)
  // (a handle over several device shards has no single row matrix: its KNN branch goes through the staged entry point, which
  // routes every label to the shard that owns it -- RSGPU_Hits_KnnRerank)
  if (want_knn && !f) tiles = general = false;
  if (f) {
    if (!knn_identity) tiles = general = false;  // (a multi-value chain over a type without a chain kernel)
    if (knn_identity) f->upload_query((tiles || general) ? ca.c : cb.c, a->query, true);
  }
  if (tiles) {
    if (hybrid_two_launches(a, f, knn_rows, want_score, want_knn, ca.c, cb.c, sc, prof, ev)) {
      tls_hybrid_path = 1;
      return 0;
    }
    a->n_hits = a->n_top = a->n_knn = 0;
    if (f) f->upload_query(cb.c, a->query, true);  // (the KNN branch of the staged pipeline reads it on its own stream)
  }
  if (general) {
    if (hybrid_general(a, a->lists, hyb_groups_flat(a->lists, a->n_lists), -1, 0, a->hits_out, f, knn_rows, want_score, want_knn,
                       ca.c, cb.c, sc, prof, ev)) {
      tls_hybrid_path = 2;
      return 0;
    }
    a->n_hits = a->n_top = a->n_knn = 0;
    if (f) f->upload_query(cb.c, a->query, true);
  }

  // ---- intersect (stream A) ----
  if (prof) HIP_CHECK(hipEventRecord(ev.e[0], ca->stream));
  uint32_t *h_total = ca->h_counters;  // pinned, device-visible
  intersect_async(h.get(), a->lists, a->n_lists, ca.c, sc, h_total);
  if (prof) HIP_CHECK(hipEventRecord(ev.e[1], ca->stream));
  // The hit count decides every later launch.  It is written into pinned memory by the scan kernel, one kernel BEFORE
  // the intersection's last (the ordered write): polling it instead of synchronising lets the host enqueue both
  // branches while that kernel still runs -- stream A orders itself, stream B waits for the event recorded here.
  HIP_CHECK(hipEventRecord(ca->ev1, ca->stream));
  if (!prof) {
    volatile uint32_t *pending = h_total;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0; *pending == kCountPending; spin++) {
      cpu_relax();
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
  }
  if (prof || h_total[0] == kCountPending) HIP_CHECK(hipStreamSynchronize(ca->stream));  // sync #1 (profiling / a very slow query)
  poll_deadline(ca.c, nullptr);
  HIP_CHECK(hipStreamWaitEvent(cb->stream, ca->ev1, 0));
  const uint32_t len = h_total[0];
  h->len = len;
  a->n_hits = len;

  // ---- branch B: ad-hoc KNN over the hits (stream B, asynchronous; enqueued FIRST -- it is the longer branch) ----
  int knn_mode = 0;       // which asynchronous select was enqueued (0: none -- the synchronous select runs after sync #2)
  uint32_t knn_k = 0;
  bool knn_on_host_map = false;
  if (want_knn && len) {
    if (knn_identity) {
      knn_k = (uint32_t)std::min<size_t>(a->k, len);
      if (prof) HIP_CHECK(hipEventRecord(ev.e[4], cb->stream));
      // Most hits of a text filter have no vector (configs[4]: one in ten): compact the ones that do -- in any order, a
      // wave-aggregated append -- so that the gather runs dense (U rows in flight per wave instead of mostly skipped
      // slots) and the selection is ONE workgroup's pass over (distance key, hit index) pairs: the hit index breaks ties
      // the way the reference does (ascending doc id), whatever order the append produced.
      const uint32_t m_up = std::min<uint32_t>(len, QueryCtx::kCandCap);
      cb->ensure_gather(std::max<uint32_t>(m_up, knn_k + 1));
      cb->ensure_out(knn_k);
      cb->h_fcnt[1] = 0;
      cb->h_fcnt[2] = 0;
      if (knn_k <= knn_topk_max_k()) {
        // three launches, no memset, no host round trip: append -> dense gather bounded by the device-side count ->
        // one top-k kernel that also fetches the winners' doc ids and re-arms the counters
        sc.knn_cnt.ensure(4);
        sc.knn_part.ensure(knn_topk_scratch_bytes() / sizeof(uint64_t));
        if (sc.knn_dirty) HIP_CHECK(hipMemsetAsync(sc.knn_cnt.p, 0, 4 * sizeof(uint32_t), cb->stream));
        sc.knn_dirty = true;
        launch_labels_to_cand(h->ids.p, len, h->base, knn_rows, cb->d_ids, cb->d_cand, sc.knn_cnt.p, QueryCtx::kCandCap, cb->stream);
        launch_gather(f->device_rows(), f->stride(), (uint32_t)f->dim, f->ktype, f->kmetric, cb->d_ids, m_up, cb->d_query,
                      cb->d_dists, cb->stream, sc.knn_cnt.p);
        launch_knn_chain_min(f->device_rows(), f->stride(), f->ktype, f->kmetric, cb->d_ids, m_up, sc.knn_cnt.p, knn_rows, cb->d_query,
                             cb->d_dists, cb->stream);
        launch_knn_topk(cb->d_dists, cb->d_cand, sc.knn_cnt.p, QueryCtx::kCandCap, knn_k, h->ids.p, sc.knn_part.p,
                        sc.knn_cnt.p + 1, cb->h_out_rows, (uint32_t *)cb->h_out_keys, cb->h_ids, cb->h_fcnt + 2,
                        cb->h_fcnt + 1, cb->stream);
        knn_mode = 2;
      } else {
      HIP_CHECK(hipMemsetAsync(cb->d_fcnt, 0, 4 * sizeof(uint32_t), cb->stream));
      HIP_CHECK(hipMemsetAsync(cb->d_ids, 0xFF, (size_t)m_up * sizeof(uint32_t), cb->stream));  // unused slots: "no vector"
      launch_labels_to_cand(h->ids.p, len, h->base, knn_rows, cb->d_ids, cb->d_cand, cb->d_fcnt, QueryCtx::kCandCap, cb->stream);
      launch_gather(f->device_rows(), f->stride(), (uint32_t)f->dim, f->ktype, f->kmetric, cb->d_ids, m_up, cb->d_query,
                    cb->d_dists, cb->stream);
      launch_knn_chain_min(f->device_rows(), f->stride(), f->ktype, f->kmetric, cb->d_ids, m_up, nullptr, knn_rows, cb->d_query,
                           cb->d_dists, cb->stream);
      launch_cand_set_keys(cb->d_cand, cb->d_dists, m_up, cb->stream);
      launch_batch_select_cand(cb->d_cand, cb->d_fcnt, QueryCtx::kCandCap, knn_k, 1, cb->h_out_rows, (uint32_t *)cb->h_out_keys,
                               cb->h_fcnt + 2, knn_k, cb->h_fcnt + 1, cb->stream);
      // doc ids of the winners: hit index -> ids[], straight behind the select (no host round trip in between)
      launch_gather_u32_counted(h->ids.p, len, cb->h_out_rows, cb->h_fcnt + 2, knn_k, cb->h_ids, cb->stream);
      knn_mode = 1;
      }
      HIP_CHECK(hipGetLastError());
      if (prof) HIP_CHECK(hipEventRecord(ev.e[5], cb->stream));
    } else {
      knn_on_host_map = true;  // general label map lives on the host: the staged entry point handles it below
    }
  }

  // ---- branch A: score + top-N prefilter (stream A, asynchronous) ----
  bool prefiltered = false, radix_topn = false;
  const uint32_t top_n = (uint32_t)std::min<size_t>(a->top_n, len);
  if (want_score && len) {
    ScoreParams P;
    bool max_norm = false;
    fill_score_params(P, h.get(), a->table, a->score, &max_norm);
    h->scores.ensure(h->cap);
    h->keys.ensure(h->cap);
    if (slop_dependent(P.scorer)) P.slops = hit_slops(h.get(), ca.c);
    prefiltered = !max_norm && top_n <= 32 && len >= (1u << 14);
    if (prefiltered) sc.skeys32.ensure(len + 4);
    launch_score(P, h->ids.p, h->freqs.p, len, h->cap, a->table->doc_len.p, a->table->doc_score.p, a->table->max_freq.p,
                 a->table->n, h->scores.p, h->keys.p, ca->stream, prefiltered ? sc.skeys32.p : nullptr);
    if (max_norm) {
      sc.maxkey.ensure(1);
      HIP_CHECK(hipMemsetAsync(sc.maxkey.p, 0, sizeof(uint64_t), ca->stream));
      launch_score_max_normalize(h->scores.p, h->keys.p, len, sc.maxkey.p, ca->stream);
    }
    h->scored = true;
    if (prof) HIP_CHECK(hipEventRecord(ev.e[2], ca->stream));
    if (prefiltered) {
      ca->ensure_out(kFetchCap);
      ca->ensure_gather(kFetchCap);
      const uint32_t per = len >= (1u << 16) ? 64 : 16;
      launch_sample_threshold(sc.skeys32.p, len, per, top_n, ca->d_tau, ca->d_fcnt, ca->stream);
      launch_filter_keys(sc.skeys32.p, len, ca->d_tau, ca->d_cand, ca->d_fcnt, QueryCtx::kCandCap, ca->stream);
      ca->h_fcnt[2] = 0;
      launch_fetch_cand64(ca->d_cand, ca->d_fcnt, kFetchCap, h->keys.p, h->ids.p, ca->h_out_rows, ca->h_out_keys,
                          ca->h_ids, ca->h_fcnt + 2, ca->stream);
    } else {
      radix_topn = true;
    }
    HIP_CHECK(hipGetLastError());
    if (prof) HIP_CHECK(hipEventRecord(ev.e[3], ca->stream));
  }

  if (want_knn && len && knn_identity) {
    HIP_CHECK(hipStreamSynchronize(cb->stream));  // sync #2a (branch B)
    if (knn_mode == 2) sc.knn_dirty = false;      // knn_topk_kernel ran to its end: the counters are back at zero
  }
  HIP_CHECK(hipStreamSynchronize(ca->stream));  // sync #2 (branch A)
  poll_deadline(nullptr, nullptr);

  // ---- results ----
  if (want_score && len) {
    std::vector<Hit> top;
    std::vector<uint32_t> top_doc;
    if (prefiltered) {
      const uint32_t n_cand = ca->h_fcnt[2];
      if (n_cand > kFetchCap || n_cand < top_n) {
        radix_topn = true;  // too many ties at the threshold for the host to settle: exact radix select on the device
      } else {
        std::vector<uint32_t> ord(n_cand);
        std::iota(ord.begin(), ord.end(), 0u);
        const uint32_t *rows = ca->h_out_rows;
        const uint64_t *keys = ca->h_out_keys;
        std::partial_sort(ord.begin(), ord.begin() + top_n, ord.end(), [&](uint32_t x, uint32_t y) {
          return keys[x] != keys[y] ? keys[x] < keys[y] : rows[x] < rows[y];
        });
        for (uint32_t i = 0; i < top_n; i++) {
          top.push_back(Hit{rows[ord[i]], keys[ord[i]]});
          top_doc.push_back(ca->h_ids[ord[i]]);
        }
      }
    }
    if (radix_topn) {
      radix_select(ca.c, h->keys.p, 8, len, top_n, Bound(), top, nullptr);
      ca->ensure_gather(top.size() + 1);
      for (size_t i = 0; i < top.size(); i++) ca->h_out_rows[i] = top[i].row;
      launch_gather_u32(h->ids.p, ca->h_out_rows, (uint32_t)top.size(), ca->h_ids, ca->stream);
      HIP_CHECK(hipStreamSynchronize(ca->stream));
      top_doc.assign(ca->h_ids, ca->h_ids + top.size());
    }
    for (size_t i = 0; i < top.size(); i++) {
      if (a->top_ids) a->top_ids[i] = h->base + top_doc[i];
      if (a->top_scores) a->top_scores[i] = key2score(top[i].key);
    }
    a->n_top = top.size();
  }
  if (want_knn && len) {
    if (knn_on_host_map) {
      if (index_lock.owns_lock()) index_lock.unlock();  // (the staged entry point takes the index's locks itself)
      long m = RSGPU_Hits_KnnRerank(h.get(), a->index, a->query, a->k, a->knn_ids, a->knn_dists);
      if (m < 0) return -1;
      a->n_knn = (size_t)m;
    } else {
      struct Win {
        uint32_t key, row, id;
      };
      std::vector<Win> win;
      uint32_t got = 0;
      if (knn_mode && !cb->h_fcnt[1]) {  // ([1]: more than kCandCap hits have a vector -- the list overflowed)
        got = std::min<uint32_t>(cb->h_fcnt[2], knn_k);
        const uint32_t *k32 = reinterpret_cast<const uint32_t *>(cb->h_out_keys);
        for (uint32_t i = 0; i < got; i++) win.push_back(Win{k32[i], cb->h_out_rows[i], cb->h_ids[i]});
      } else {  // the general form: every hit keeps its slot (absent rows -> NaN), keys selected by (key, hit index)
        sc.rows.ensure(len);
        sc.dists.ensure(len);
        sc.keys32.ensure(len);
        launch_labels_to_rows(h->ids.p, len, h->base, knn_rows, sc.rows.p, cb->stream);
        launch_gather(f->device_rows(), f->stride(), (uint32_t)f->dim, f->ktype, f->kmetric, sc.rows.p, len, cb->d_query,
                      sc.dists.p, cb->stream);
        launch_knn_chain_min(f->device_rows(), f->stride(), f->ktype, f->kmetric, sc.rows.p, len, nullptr, knn_rows, cb->d_query,
                             sc.dists.p, cb->stream);
        launch_dist_to_keys(sc.dists.p, len, sc.keys32.p, cb->stream);
        std::vector<Hit> knn_hits;
        select_keys32(cb.c, sc.keys32.p, len, knn_k, knn_hits);
        cb->ensure_gather(knn_hits.size() + 1);
        for (size_t i = 0; i < knn_hits.size(); i++) cb->h_out_rows[i] = knn_hits[i].row;
        launch_gather_u32(h->ids.p, cb->h_out_rows, (uint32_t)knn_hits.size(), cb->h_ids, cb->stream);
        HIP_CHECK(hipStreamSynchronize(cb->stream));
        for (size_t i = 0; i < knn_hits.size(); i++) win.push_back(Win{(uint32_t)knn_hits[i].key, knn_hits[i].row, cb->h_ids[i]});
      }
      std::sort(win.begin(), win.end(), [](const Win &x, const Win &y) { return x.key != y.key ? x.key < y.key : x.row < y.row; });
      size_t out = 0;
      for (const Win &w : win) {
        if (w.key == 0xFFFFFFFFu) continue;  // NaN: the doc has no vector (hybrid_reader.c:317-320)
        if (a->knn_ids) a->knn_ids[out] = h->base + w.id;
        if (a->knn_dists) a->knn_dists[out] = (double)key_to_dist(w.key);
        out++;
      }
      a->n_knn = out;
    }
  }
  if (prof) {
    float ms = 0;
    prof_ms[0] = 0;
    if (hipEventElapsedTime(&ms, ev.e[0], ev.e[1]) == hipSuccess) prof_ms[1] = ms;
    if (want_score && len) {
      if (hipEventElapsedTime(&ms, ev.e[1], ev.e[2]) == hipSuccess) prof_ms[2] = ms;
      if (hipEventElapsedTime(&ms, ev.e[2], ev.e[3]) == hipSuccess) prof_ms[3] = ms;
    }
    if (want_knn && len && !knn_on_host_map && hipEventSynchronize(ev.e[5]) == hipSuccess &&
        hipEventElapsedTime(&ms, ev.e[4], ev.e[5]) == hipSuccess)
      prof_ms[4] = ms;
  }
  if (a->hits_out) *a->hits_out = h.release();
  return 0;
  S_CATCH_HYBRID(a)
}

/* RSGPU_HybridQuery over a two-level query tree (include/rsgpu_search.h): the general tile kernel when the root is an
 * intersection of at most eight lists with a term to drive it, else stage by stage -- RSGPU_EvalTree, then the entry points a
 * caller would use on its hit list.  Same answers. */
extern "C" int RSGPU_HybridTreeQuery(const RSGPU_TreeQuery *q, RSGPU_HybridQueryArgs *a) {
  if (!q || !a || !q->lists || !q->n_groups || !q->group_first) {
    last_error() = "RSGPU_HybridTreeQuery: empty tree";
    return -1;
  }
  S_TRY
  const size_t n_lists = q->group_first[q->n_groups];
  if (q->n_groups > (size_t)kMaxLists || n_lists > (size_t)kMaxLists || !n_lists)
    throw std::runtime_error("RSGPU_HybridTreeQuery: at most 32 groups and 32 terms");
  if (q->root_op != RSGPU_OP_INTERSECT && q->root_op != RSGPU_OP_UNION) throw std::runtime_error("RSGPU_HybridTreeQuery: bad root_op");
  HybridPlan plan("RSGPU_HybridTreeQuery", a, q->lists, n_lists);
  const bool want_score = plan.want_score, want_knn = plan.want_knn;
  FlatIndex *f = plan.f;

  // A root UNION of terms / intersections of terms (`a | b`, `(a b) | (c d)`: round 5) takes the tile kernel too -- one pass per
  // child, one reduce -- when nobody asked for the hit list and the scorer does not divide by the result's slop (a union result
  // holds the matched children only: its slop differs from hit to hit)
  bool root_union = q->root_op == RSGPU_OP_UNION && !(want_score && slop_dependent(a->score->scorer));
  if (root_union && q->group_op)
    for (size_t g = 0; g < q->n_groups && root_union; g++)
      root_union = q->group_op[g] == RSGPU_OP_TERM || q->group_op[g] == RSGPU_OP_INTERSECT;
  bool general = scan_tuning().hybrid_tiles && scan_tuning().hybrid_tree_tiles && (want_score || want_knn) &&
                 (q->root_op == RSGPU_OP_INTERSECT || root_union) && n_lists <= (size_t)kHybTreeMaxLists && (!want_knn || (f && f->key_bytes == 4));
  root_union = root_union && general;
  const bool norm = want_score && a->score->scorer == RSGPU_SCORER_BM25STD_NORM;
  // (an EXCLUDED list may be empty -- `a -b` with an empty b is `a`: the list is simply not probed -- a required one may not)
  std::vector<char> excluded_list(n_lists, 0);
  if (q->group_op)
    for (size_t g = 0; g < q->n_groups; g++)
      if (q->group_op[g] == RSGPU_OP_NOT)
        for (size_t l = q->group_first[g]; l < q->group_first[g + 1] && l < n_lists; l++) excluded_list[l] = 1;
  for (size_t l = 0; l < n_lists && general; l++) general = excluded_list[l] || q->lists[l]->n_entries > 0;
  if (general) {
    const std::vector<HybGroup> groups = hyb_groups_tree(q, n_lists);
    const uint32_t tiles = root_union ? hyb_union_tiles(groups, q->lists) : hyb_intersection_tiles(groups, q->lists, a->hits_out != nullptr);
    general = tiles > 0 &&
              hybrid_tree_supported(f ? f->ktype : 0, f ? f->kmetric : 0, f ? (uint32_t)(f->stride() / 16) : 1u, tiles,
                                    want_score ? (uint32_t)a->top_n + (norm ? 1u : 0u) : 0u, want_knn ? (uint32_t)a->k : 0u, (int)n_lists);
    if (general) {
      HybridTileRun run(plan);
      general = !f || run.labels_ok;
      if (general && f) f->upload_query(run.ca.c, a->query, true);
      if (general && hybrid_general(a, q->lists, groups, q->max_slop, q->in_order, a->hits_out, f, run.knn_rows, want_score, want_knn,
                                    run.ca.c, run.cb.c, run.sc, run.prof, run.ev, root_union)) {
        tls_hybrid_path = 2;
        return 0;
      }
      a->n_hits = a->n_top = a->n_knn = 0;
    }
  }
  if (q->group_op)
    for (size_t g = 0; g < q->n_groups; g++)
      if (q->group_op[g] == RSGPU_OP_NOT)
        throw std::runtime_error("RSGPU_HybridTreeQuery: a query with NOT children runs on the general tile kernel only -- a root "
                                 "intersection of at most eight lists with a term or a union of terms to drive it, top_n / k <= 64, labels a "
                                 "device table holds (RSGPU_FlatIndex_LabelTable != 2)");
  // stage by stage (the index lock is released: the entry points below take it themselves)
  // (RSGPU_EvalTree may build the list with the tile kernel; this QUERY runs stage by stage: the plan says path 0)
  return plan.staged(std::unique_ptr<RSGPU_Hits>(RSGPU_EvalTree(q)));
  S_CATCH_HYBRID(a)
}

// ---- RSGPU_HybridTreeNodesQuery: the hybrid query over a tree of any depth ----
// The node array as a tree, every intersection's children in the order it iterates them (ascending estimate x sort weight,
// stable: intersection.rs:94-119; a union keeps the query's order) -- the order RSGPU_EvalTreeNodes gives the result tree.
namespace {
struct QNode {
  int op = 0, list = -1;  // 0 term, 1 union, 2 intersection, 3 not (its children: the excluded terms)
  double weight = 1.0;
  std::vector<int> kids;
  size_t estimate = 0;
  double key = 0.0;
  int leaf_first = 0, n_leaves = 0;  // its terms among the leaves of the root child it belongs to
  bool has_union = false;            // some aggregate below (or itself) is a union
};
struct QTree {
  std::vector<QNode> n;
  int root = -1, depth = 0;
  bool windows = false;   // some node below the root carries max_slop / in_order
  bool has_not = false;   // some node is a NOT (no staged form: RSGPU_EvalTreeNodes has no such node)
  long root_slop = -1;    // the root's own window
  int root_in_order = 0;
};
// false: not a well-formed post-order array (RSGPU_EvalTreeNodes names the fault)
bool parse_nodes(const RSGPU_TreeNode *nodes, size_t n_nodes, RSGPU_Postings *const *lists, size_t n_lists, QTree &t) {
  if (n_nodes > (size_t)kMaxNodes || n_lists > (size_t)kMaxLists) return false;
  std::vector<int> st;
  std::vector<char> used(n_lists, 0);
  for (size_t i = 0; i < n_nodes; i++) {
    const RSGPU_TreeNode &nd = nodes[i];
    QNode q;
    if (nd.op == RSGPU_OP_TERM) {
      if (nd.list >= n_lists || used[nd.list]) return false;
      used[nd.list] = 1;
      q.list = (int)nd.list;
      q.estimate = lists[nd.list]->n_entries;
      q.key = intersection_sort_key(q.estimate, 0, 1);
    } else if (nd.op == RSGPU_OP_UNION || nd.op == RSGPU_OP_INTERSECT) {
      if (!nd.n_children || nd.n_children > st.size()) return false;
      q.op = nd.op == RSGPU_OP_UNION ? 1 : 2;
      q.weight = nd.weight;
      q.kids.assign(st.end() - (long)nd.n_children, st.end());
      st.resize(st.size() - nd.n_children);
      const bool window = nd.op == RSGPU_OP_INTERSECT && (nd.max_slop >= 0 || nd.in_order);
      if (window && i + 1 < n_nodes) t.windows = true;
      if (window && i + 1 == n_nodes) {
        t.root_slop = nd.max_slop;
        t.root_in_order = nd.in_order ? 1 : 0;
      }
      if (q.op == 2) {
        if (!nd.in_order)  // (in_order: the query's order is the order the children must appear in)
          std::stable_sort(q.kids.begin(), q.kids.end(), [&](int x, int y) { return t.n[x].key < t.n[y].key; });
        q.estimate = ~(size_t)0;
        for (int k : q.kids) q.estimate = std::min(q.estimate, t.n[k].estimate);
      } else {
        for (int k : q.kids) q.estimate += t.n[k].estimate;
      }
      q.has_union = q.op == 1;
      for (int k : q.kids) q.has_union = q.has_union || t.n[k].has_union;
      q.key = intersection_sort_key(q.estimate, q.op, nd.n_children);
    } else if (nd.op == RSGPU_OP_NOT) {  // (RSGPU_HybridTreeNodesQuery only: a child of the root intersection over terms)
      if (!nd.n_children || nd.n_children > st.size()) return false;
      q.op = 3;
      q.weight = nd.weight;
      q.kids.assign(st.end() - (long)nd.n_children, st.end());
      st.resize(st.size() - nd.n_children);
      for (int k : q.kids)
        if (t.n[k].op != 0) return false;
      q.estimate = ~(size_t)0;  // (a Not's estimate is max_doc_id: it sorts behind every real child)
      q.key = 1.0e300;
      t.has_not = true;
    } else {
      return false;
    }
    t.n.push_back(q);
    st.push_back((int)i);
  }
  if (st.size() != 1) return false;
  t.root = st[0];
  return true;
}
// one child of the root as a HybGroup: its leaves in result order, its result tree, what a hit must hold of it.
// false: a shape the tile kernel's predicate (sets of which one / all must match) cannot express -- a union below a nested
// intersection below a union
struct GroupBuilder {
  QTree &t;
  HybGroup &g;
  int depth_max = 0;
  bool misplaced_not = false;
  void emit(int i, int depth) {  // leaves, tree nodes (post-order)
    QNode &q = t.n[i];
    depth_max = std::max(depth_max, depth);
    if (q.op == 3) misplaced_not = true;
    q.leaf_first = (int)g.lists.size();
    if (q.op == 0) {
      g.tree.push_back(TNode{0, (uint8_t)g.lists.size(), 0, 1.0});
      g.lists.push_back(q.list);
    } else {
      for (int k : q.kids) emit(k, depth + 1);
      g.tree.push_back(TNode{(uint8_t)q.op, 0, (uint16_t)q.kids.size(), q.weight});
    }
    q.n_leaves = (int)g.lists.size() - q.leaf_first;
  }
  uint32_t mask(int i) const { return (uint32_t)(((1ull << t.n[i].n_leaves) - 1ull) << t.n[i].leaf_first); }
  bool under_union(int i) {  // a child of a union, or of a union below a union
    const QNode &q = t.n[i];
    if (q.op == 0) return true;
    if (q.op == 2) {
      if (q.has_union) return false;
      g.whole.push_back(mask(i));
      return true;
    }
    for (int k : q.kids)
      if (!under_union(k)) return false;
    return true;
  }
  void must_only(int i) {  // the terms every hit holds: those below intersections only
    const QNode &q = t.n[i];
    if (q.op == 0) g.must.push_back(mask(i));
    else if (q.op == 2)
      for (int k : q.kids) must_only(k);
  }
  bool required(int i) {  // a node every hit holds
    const QNode &q = t.n[i];
    if (q.op == 0) {
      g.any_of.push_back(mask(i));
      g.must.push_back(mask(i));
      return true;
    }
    if (q.op == 2) {
      for (int k : q.kids)
        if (!required(k)) return false;
      return true;
    }
    for (int k : q.kids)
      if (!under_union(k)) return false;
    g.any_of.push_back(mask(i));
    return true;
  }
};
// the root's children in the order the root intersection iterates them; false: no form on the tile kernel
bool tree_groups(QTree &t, std::vector<HybGroup> &groups) {
  const QNode &root = t.n[t.root];
  if (root.op != 2 || t.windows) return false;
  bool deep = false;
  for (int c : root.kids) {  // (already in iteration order: parse_nodes sorted them)
    const QNode &q = t.n[c];
    HybGroup g;
    if (q.op == 3) {  // excluded terms: no leaf of the result tree, a virtual child of frequency 0 (hyb_groups_tree)
      g.op = 3;
      g.weight = q.weight;
      g.estimate = ~(size_t)0;
      for (int k : q.kids) g.lists.push_back(t.n[k].list);
      groups.push_back(std::move(g));
      continue;
    }
    g.op = q.op;
    g.weight = q.op ? q.weight : 1.0;  // (a term's own weight stays in RSGPU_ScoreArgs.weight)
    g.estimate = q.estimate;
    bool plain = true;  // a term, or an aggregate of terms: the two-level forms hybrid_general knows
    for (int k : q.kids) plain = plain && t.n[k].op == 0;
    GroupBuilder b{t, g};
    b.emit(c, 1);
    if (b.misplaced_not || b.depth_max > kHybDeepLevels) return false;
    t.depth = std::max(t.depth, b.depth_max);
    if (plain) {
      g.tree.clear();
    } else {
      g.deep = deep = true;
      g.n_children = q.kids.size();
      if (!b.required(c)) {  // (a union below an intersection below a union: the match folded over the tree, HybridTreeArgs::tree_pred)
        g.any_of.clear();
        g.whole.clear();
        g.must.clear();
        g.pred_tree = true;
        b.must_only(c);
      }
    }
    groups.push_back(std::move(g));
  }
  // (the root's own window over nested children -- round 6: a child's offsets are those of its leaves in the result, merged
  // (RSIndexResult_IterateOffsets of an aggregate, src/index_result/index_result.c:51-103 reads them through it): the two-level
  // proximity code over the children's leaf ranges, absent leaves skipped)
  t.depth = deep ? std::max(t.depth, 3) : t.depth;
  return true;
}
}  // namespace

/* RSGPU_HybridQuery over a query tree of any depth (include/rsgpu_search.h): the general tile kernel for a root intersection over
 * at most eight lists with a term to drive it, whose nested aggregates the kernel's predicate expresses (tree_groups) -- every
 * list probed in place, the score folded over the whole result tree (score_one<true>) -- else stage by stage: RSGPU_EvalTreeNodes,
 * then the entry points a caller would use on its hit list.  Same answers. */
extern "C" int RSGPU_HybridTreeNodesQuery(const RSGPU_TreeNode *nodes, size_t n_nodes, RSGPU_HybridQueryArgs *a) {
  if (!nodes || !n_nodes || !a || !a->lists || !a->n_lists) {
    last_error() = "RSGPU_HybridTreeNodesQuery: empty tree";
    return -1;
  }
  S_TRY
  if (n_nodes > (size_t)kMaxNodes || a->n_lists > (size_t)kMaxLists) throw std::runtime_error("RSGPU_HybridTreeNodesQuery: at most 64 nodes over 32 terms");
  const size_t n_lists = a->n_lists;
  HybridPlan plan("RSGPU_HybridTreeNodesQuery", a, a->lists, n_lists);
  const bool want_score = plan.want_score, want_knn = plan.want_knn;
  FlatIndex *f = plan.f;

  QTree qt;
  std::vector<HybGroup> groups;
  const bool parsed = parse_nodes(nodes, n_nodes, a->lists, n_lists, qt);
  bool general = scan_tuning().hybrid_tiles && scan_tuning().hybrid_tree_tiles && (want_score || want_knn) &&
                 n_lists <= (size_t)kHybTreeMaxLists && (!want_knn || (f && f->key_bytes == 4)) && parsed && tree_groups(qt, groups);
  bool deep = false;
  if (general) {
    // every list a leaf or an excluded term (a list no node names: RSGPU_EvalTreeNodes ignores it, the kernel's arrays would
    // not); an EXCLUDED list may be empty -- it is simply not probed -- a required one may not
    std::vector<char> seen(n_lists, 0);
    bool offsets = false;
    for (const HybGroup &g : groups) {
      deep = deep || g.deep;
      for (int li : g.lists) {
        seen[li] = 1;
        if (g.op != 3) {
          general = general && a->lists[li]->n_entries > 0;
          offsets = offsets || a->lists[li]->has_offsets();
        }
      }
    }
    for (size_t l = 0; l < n_lists; l++) general = general && seen[l];
    (void)offsets;  // (round 6: a scorer that divides by the result's slop over nested children runs on the tile path too)
  }
  if (general) {
    const bool norm = want_score && a->score->scorer == RSGPU_SCORER_BM25STD_NORM;
    const uint32_t tiles = hyb_intersection_tiles(groups, a->lists, false);
    general = tiles > 0 &&
              hybrid_tree_supported(f ? f->ktype : 0, f ? f->kmetric : 0, f ? (uint32_t)(f->stride() / 16) : 1u, tiles,
                                    want_score ? (uint32_t)a->top_n + (norm ? 1u : 0u) : 0u, want_knn ? (uint32_t)a->k : 0u, (int)n_lists);
    if (general) {
      HybridTileRun run(plan);
      general = !f || run.labels_ok;
      if (general && f) f->upload_query(run.ca.c, a->query, true);
      if (general && hybrid_general(a, a->lists, groups, qt.root_slop, qt.root_in_order, a->hits_out, f, run.knn_rows,
                                    want_score, want_knn, run.ca.c, run.cb.c, run.sc, run.prof, run.ev)) {
        tls_hybrid_path = 2;
        return 0;
      }
      a->n_hits = a->n_top = a->n_knn = 0;
    }
  }
  if (parsed && qt.has_not)
    throw std::runtime_error("RSGPU_HybridTreeNodesQuery: a query with NOT children runs on the general tile kernel only -- NOT nodes over "
                             "terms under a root intersection of at most eight lists with a term or a union of terms to drive it, "
                             "top_n / k <= 64, labels a device table holds");
  // stage by stage (the index lock is released: the entry points below take it themselves)
  return plan.staged(std::unique_ptr<RSGPU_Hits>(RSGPU_EvalTreeNodes(nodes, n_nodes, a->lists, n_lists)));
  S_CATCH_HYBRID(a)
}

