// sharded_index.cpp -- one FLAT index over several GPUs of ONE process (include/rsgpu_ext.h RSGPU_ShardedIndex_*).
//
// A Redis module is one process; the reference scales a KNN query by fanning it out to per-shard indexes and merging the
// per-shard top-K lists in a heap (coordinator, reference src/module.c:3541-3547).  This is the in-process
// form of that: the corpus is row-partitioned over N FlatIndex shards, each resident in the HBM of its own device (or
// several on one device -- the 1-GPU tests do that), every shard has a worker thread bound to its device, a query is
// posted to all workers at once, each runs the ordinary single-device scan + select whose last kernel already writes
// its K winners into pinned host memory (the buffer every device can reach), and the caller K-way merges the N sorted
// lists by (score, label).  N*K is a few dozen pairs: the merge costs less than one kernel launch, so it stays on the
// host; no data-path collective is needed inside one process.  (Between PROCESSES -- one rank per GPU under
// torch.distributed, bench.py under torchrun -- the same per-shard lists travel by one RCCL all-gather over xGMI.)
//
// REPLICA mode (SURVEY.md 8e "non-sharded alternative"): every shard holds the whole corpus, a query goes to ONE
// replica picked round-robin on the calling thread -- throughput scales with concurrent callers, latency does not change.
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include "rsgpu_ext.h"
#include "sharded_index.hpp"

using namespace rsgpu;

namespace {

// what a caller waits on: the tasks it posted (one per shard, at most), counted down by the workers
struct Completion {
  std::mutex mu;
  std::condition_variable cv;
  int remaining = 0;
  std::string error;  // first error of a closure task
};
// one unit of work for a shard's worker: a closure, or a top-k query (kept apart so that the worker can put the queries
// of several concurrent callers into ONE pass over its rows, FlatIndex::topk_pass)
struct Task {
  std::function<void()> fn;
  TopkJob *topk = nullptr;
  Completion *done = nullptr;
};

struct Shard {
  FlatIndex *flat = nullptr;
  int device = 0;
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Task> q;  // FIFO; any number of callers post concurrently
  bool stop = false;
  uint32_t last_b = 1;  // queries in the worker's previous pass (how many to wait a moment for, as in FlatIndex::topk)
};

}  // namespace

struct RSGPU_ShardedIndex {
  std::vector<std::unique_ptr<Shard>> shards;
  bool replicas = false;
  bool multi = false;
  void *log_ctx = nullptr;
  std::atomic<uint64_t> rr{0};
  std::atomic<int> last_mode{EMPTY_MODE};
  std::vector<VecSimIndex> handles;  // per-shard ABI handles (RSGPU_ShardedIndex_Shard; per-shard iterators / contexts)
  // exchange statistics of the fan-out queries (bench.py's `collective` record): nanoseconds between the moment the last
  // shard's winners are in host memory and the merged reply
  std::atomic<uint64_t> merges{0}, merge_ns{0};
  // "shard_exchange" = 1: the per-shard top-k travel through ONE ncclAllGather + a merge kernel (shard_comm.cpp) instead of
  // the host merge -- one communicator per shard device, created by the first such query; collectives are ordered, so
  // these queries go one at a time
  std::mutex exchange_mu;
  std::vector<RSGPU_ShardComm *> comms;
  std::atomic<uint64_t> rccl_queries{0}, rccl_ns{0};
  size_t n() const { return shards.size(); }
  Shard *pick() { return shards[rr++ % shards.size()].get(); }  // replica mode: round-robin
};

static void task_done(Completion *c, const std::string &err) {
  std::lock_guard<std::mutex> g(c->mu);  // (notify under the lock: the waiter owns the object and may destroy it next)
  if (!err.empty() && c->error.empty()) c->error = err;
  if (--c->remaining == 0) c->cv.notify_all();
}

// A shard's worker: its device is current here for good.  Top-k queries at the head of the queue -- the fan-outs of
// concurrent callers -- are answered together, up to kMqMaxQueries per pass over the shard's rows (the coalescer of
// FlatIndex::topk, moved to where the queries of a sharded handle meet); everything else runs task by task.
static void worker_main(Shard *s) {
  (void)hipSetDevice(s->device);
  for (;;) {
    std::vector<Task> batch;
    {
      std::unique_lock<std::mutex> g(s->mu);
      s->cv.wait(g, [&] { return s->stop || !s->q.empty(); });
      if (s->q.empty()) return;  // (stop: the queue is drained first)
      auto leading = [&] {
        size_t c = 0;
        while (c < s->q.size() && s->q[c].topk) c++;
        return c;
      };
      // (the "coalesce" / "coalesce_min_mib" knobs and the multi-query scan's shapes gate the batching here as they do in
      // FlatIndex::topk: a shard they rule out answers one task at a time, without the wait)
      if (s->q.front().topk && !s->flat->coalescible(s->q.front().topk->k)) {
        batch.push_back(std::move(s->q.front()));
        s->q.pop_front();
      } else if (s->q.front().topk) {
        const size_t expect = std::min<uint32_t>(s->last_b, kMqMaxQueries);
        if (leading() < expect && !s->stop) {
          const int us = s->flat->coalesce_linger_us();
          if (us > 0) s->cv.wait_for(g, std::chrono::microseconds(us), [&] { return s->stop || leading() >= expect; });
        }
        const size_t take = std::min<size_t>(leading(), kMqMaxQueries);
        for (size_t i = 0; i < take; i++) {
          batch.push_back(std::move(s->q.front()));
          s->q.pop_front();
        }
      } else {
        batch.push_back(std::move(s->q.front()));
        s->q.pop_front();
      }
    }
    if (batch[0].topk) {
      TopkJob *jobs[kMqMaxQueries];
      for (size_t i = 0; i < batch.size(); i++) jobs[i] = batch[i].topk;
      std::exception_ptr err;
      try {
        s->flat->topk_pass(jobs, batch.size());
      } catch (...) {
        err = std::current_exception();
      }
      {  // the callers there are = the ones answered just now + the ones queued meanwhile (see FlatIndex::topk)
        std::lock_guard<std::mutex> g(s->mu);
        size_t queued = 0;
        while (queued < s->q.size() && s->q[queued].topk) queued++;
        s->last_b = (uint32_t)std::min<size_t>(batch.size() + queued, kMqMaxQueries);
      }
      for (Task &t : batch) {
        if (err && !t.topk->reply) t.topk->err = err;
        task_done(t.done, std::string());
      }
    } else {
      std::string err;
      try {
        if (batch[0].fn) batch[0].fn();
      } catch (const std::exception &e) {
        err = e.what();
      } catch (...) {
        err = "unknown error";
      }
      task_done(batch[0].done, err);
    }
  }
}

static void post(Shard *s, Task &&t) {
  {
    std::lock_guard<std::mutex> g(s->mu);
    s->q.push_back(std::move(t));
  }
  s->cv.notify_one();
}
static void wait_for(Completion &c) {
  std::unique_lock<std::mutex> g(c.mu);
  c.cv.wait(g, [&] { return c.remaining == 0; });
}

// Runs jobs[i] on shard i's worker (its device is current there), all at once; returns when every one has finished.
// Throws the first error.  Any number of callers may be in here at the same time: the workers queue.
static void run_on_shards(RSGPU_ShardedIndex *si, const std::vector<std::function<void()>> &jobs) {
  Completion c;
  for (size_t i = 0; i < si->n(); i++) c.remaining += jobs[i] ? 1 : 0;
  if (!c.remaining) return;
  for (size_t i = 0; i < si->n(); i++)
    if (jobs[i]) post(si->shards[i].get(), Task{jobs[i], nullptr, &c});
  wait_for(c);
  if (!c.error.empty()) throw std::runtime_error(c.error);
}

// shard code that runs on the CALLER's thread selects the shard's device there: put the caller's device back afterwards
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

static bool by_score_then_id(const VecSimQueryResult &a, const VecSimQueryResult &b) {
  return score_id_before(a.score, a.id, b.score, b.id);  // (the order the shards' replies are in: NaN scores last)
}

// K-way merge of per-shard replies (each ascending by (score, id)); frees them.
static VecSimQueryReply *merge_replies(std::vector<VecSimQueryReply *> &replies, size_t k, bool cut,
                                       VecSimQueryReply_Order order) {
  size_t total = 0;
  bool timed_out_any = false;
  for (VecSimQueryReply *r : replies)
    if (r) {
      total += r->len;
      timed_out_any |= r->code == VecSim_QueryReply_TimedOut;
    }
  VecSimQueryReply *out = nullptr;
  if (timed_out_any) {
    out = new_reply(0, VecSim_QueryReply_TimedOut);
  } else {
    const size_t take = cut ? std::min(k, total) : total;
    out = new_reply(take, VecSim_QueryReply_OK);
    std::vector<size_t> pos(replies.size(), 0);
    for (size_t n = 0; n < take; n++) {
      int best = -1;
      for (size_t i = 0; i < replies.size(); i++) {
        VecSimQueryReply *r = replies[i];
        if (!r || pos[i] >= r->len) continue;
        if (best < 0 || by_score_then_id(r->results[pos[i]], replies[best]->results[pos[best]])) best = (int)i;
      }
      out->results[n] = replies[best]->results[pos[best]++];
    }
    if (order == BY_ID)
      std::sort(out->results, out->results + out->len,
                [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return a.id < b.id; });
  }
  for (VecSimQueryReply *&r : replies) {
    VecSimQueryReply_Free(r);
    r = nullptr;
  }
  return out;
}

namespace rsgpu {

void sharded_free(RSGPU_ShardedIndex *si);

RSGPU_ShardedIndex *sharded_new(const BFParams &p, void *log_ctx, int n_shards, const int *devices, bool replicas) {
  if (n_shards < 1 || n_shards > 64) throw std::runtime_error("1..64 shards");
  std::string why;
  if (!device_available(&why)) throw std::runtime_error(why);
  int ndev = 0;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  int prev = 0;
  HIP_CHECK(hipGetDevice(&prev));
  std::unique_ptr<RSGPU_ShardedIndex> si(new RSGPU_ShardedIndex());
  si->replicas = replicas;
  si->multi = p.multi;
  si->log_ctx = log_ctx;
  try {
    for (int i = 0; i < n_shards; i++) {
      const int dev = devices ? devices[i] : i % ndev;
      if (dev < 0 || dev >= ndev) throw std::runtime_error("no such device");
      HIP_CHECK(hipSetDevice(dev));
      std::unique_ptr<Shard> s(new Shard());
      s->device = dev;
      s->flat = new FlatIndex(p, log_ctx);
      si->shards.push_back(std::move(s));
    }
  } catch (...) {
    for (auto &s : si->shards) delete s->flat;
    (void)hipSetDevice(prev);
    throw;
  }
  HIP_CHECK(hipSetDevice(prev));
  try {
    si->handles.resize(si->n());
    for (size_t i = 0; i < si->n(); i++) {
      si->handles[i].flat = si->shards[i]->flat;
      if (!si->replicas && si->n() > 1) si->shards[i]->worker = std::thread(worker_main, si->shards[i].get());
    }
  } catch (...) {  // (a later std::thread failed to start: the started ones are joinable and own nothing yet)
    sharded_free(si.release());
    throw;
  }
  return si.release();
}

void sharded_free(RSGPU_ShardedIndex *si) {
  if (!si) return;
  for (auto &s : si->shards) {
    if (s->worker.joinable()) {
      {
        std::lock_guard<std::mutex> g(s->mu);
        s->stop = true;
      }
      s->cv.notify_all();
      s->worker.join();
    }
    try {
      delete s->flat;
    } catch (...) {
    }
  }
  for (RSGPU_ShardComm *c : si->comms) RSGPU_ShardComm_Free(c);
  delete si;
}

void *sharded_log_ctx(RSGPU_ShardedIndex *si) { return si->log_ctx; }
FlatIndex *sharded_first(RSGPU_ShardedIndex *si) { return si->shards[0]->flat; }
int sharded_last_mode(RSGPU_ShardedIndex *si) { return si->last_mode.load(); }

size_t sharded_size(RSGPU_ShardedIndex *si) {
  if (si->replicas) return si->shards[0]->flat->size();
  size_t n = 0;
  for (auto &s : si->shards) n += s->flat->size();
  return n;
}
size_t sharded_label_count(RSGPU_ShardedIndex *si) {
  if (si->replicas) return si->shards[0]->flat->label_count();
  size_t n = 0;  // a label lives on exactly one shard
  for (auto &s : si->shards) n += s->flat->label_count();
  return n;
}
size_t sharded_memory(RSGPU_ShardedIndex *si) {
  size_t n = 0;
  for (auto &s : si->shards) n += s->flat->memory();
  return n;
}

// A label lives on exactly one shard: an existing label is overwritten (single-value) or extended (multi-value) where
// it is, a new one goes to the emptiest shard.  Replicas: every shard gets the vector.
int sharded_add(RSGPU_ShardedIndex *si, const void *blob, size_t label) {
  DeviceGuard dg;
  if (si->replicas) {
    int r = 0;
    for (auto &s : si->shards) r = s->flat->add(blob, label);
    return r;
  }
  Shard *target = nullptr;
  for (auto &s : si->shards)
    if (s->flat->contains(label)) { target = s.get(); break; }
  if (!target) {  // (a handle always has at least one shard)
    target = si->shards[0].get();
    size_t best = target->flat->size();
    for (auto &s : si->shards) {
      const size_t n = s->flat->size();
      if (n < best) { best = n; target = s.get(); }
    }
  }
  return target->flat->add(blob, label);
}

int sharded_remove(RSGPU_ShardedIndex *si, size_t label) {
  DeviceGuard dg;
  int removed = 0;
  for (auto &s : si->shards) {
    const int r = s->flat->remove(label);
    removed = si->replicas ? r : removed + r;
  }
  return removed;
}

double sharded_distance_from(RSGPU_ShardedIndex *si, size_t label, const void *normalized_blob) {
  DeviceGuard dg;
  if (si->replicas) return si->pick()->flat->distance_from(label, normalized_blob);
  for (auto &s : si->shards) {
    if (!s->flat->contains(label)) continue;
    return s->flat->distance_from(label, normalized_blob);
  }
  return NAN;
}

// The exchange over RCCL (knob "shard_exchange" = 1; BASELINE north star: "RCCL all-gather of per-shard top-K over xGMI"):
// the shards answer as they always do -- one top-k task per shard worker -- and the CALLER's thread then runs the exchange
// for all ranks (the single-thread form of a NCCL program: every rank's ncclAllGather inside one group), the merge kernel
// leaves the global k best in pinned host memory.  Communicators: one per shard device, created by the first such query.
// Collectives are ordered, so these queries pass the exchange one at a time.
static VecSimQueryReply *sharded_topk_rccl(RSGPU_ShardedIndex *si, const void *query, size_t k, VecSimQueryParams *qp,
                                           VecSimQueryReply_Order order) {
  const size_t n = si->n();
  std::vector<VecSimQueryReply *> replies(n, nullptr);
  struct FreeReplies {
    std::vector<VecSimQueryReply *> &r;
    ~FreeReplies() {
      for (VecSimQueryReply *x : r) VecSimQueryReply_Free(x);
    }
  } free_replies{replies};
  if (n == 1) {
    replies[0] = si->shards[0]->flat->topk(query, k, qp, BY_SCORE);
  } else {
    std::vector<TopkJob> jobs(n, TopkJob{query, k, qp ? qp->timeoutCtx : nullptr, BY_SCORE});
    Completion c;
    c.remaining = (int)n;
    for (size_t i = 0; i < n; i++) {
      si->shards[i]->flat->last_mode = STANDARD_KNN;
      post(si->shards[i].get(), Task{nullptr, &jobs[i], &c});
    }
    wait_for(c);
    std::exception_ptr err;
    for (size_t i = 0; i < n; i++) {
      replies[i] = jobs[i].reply;
      if (jobs[i].err && !err) err = jobs[i].err;
    }
    if (err) std::rethrow_exception(err);
  }
  for (size_t i = 0; i < n; i++)
    if (replies[i] && replies[i]->code == VecSim_QueryReply_TimedOut) return new_reply(0, VecSim_QueryReply_TimedOut);
  std::lock_guard<std::mutex> g(si->exchange_mu);
  const auto t0 = std::chrono::steady_clock::now();
  if (si->comms.empty()) {
    std::vector<int> devs;
    for (size_t i = 0; i < n; i++) {
      for (size_t j = 0; j < i; j++)
        if (si->shards[i]->device == si->shards[j]->device)
          throw std::runtime_error("shard_exchange = 1: a RCCL communicator needs one device per rank (shards " + std::to_string(j) + " and " +
                                   std::to_string(i) + " share device " + std::to_string(si->shards[i]->device) + ")");
      devs.push_back(si->shards[i]->device);
    }
    si->comms = shard_comm_init_group(devs);
  }
  std::vector<const VecSimQueryResult *> local(n);
  std::vector<size_t> n_local(n);
  for (size_t i = 0; i < n; i++) {
    local[i] = replies[i] ? replies[i]->results : nullptr;
    n_local[i] = replies[i] ? replies[i]->len : 0;
  }
  std::vector<uint64_t> labels(k);
  std::vector<double> scores(k);
  const size_t got = shard_comm_exchange_group(si->comms, local.data(), n_local.data(), k, labels.data(), scores.data());
  VecSimQueryReply *out = new_reply(got, VecSim_QueryReply_OK);
  for (size_t j = 0; j < got; j++) out->results[j] = VecSimQueryResult{(size_t)labels[j], scores[j]};
  if (order == BY_ID)
    std::sort(out->results, out->results + out->len, [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return a.id < b.id; });
  si->rccl_queries++;
  si->rccl_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return out;
}

VecSimQueryReply *sharded_topk(RSGPU_ShardedIndex *si, const void *query, size_t k, VecSimQueryParams *qp,
                               VecSimQueryReply_Order order) {
  DeviceGuard dg;
  si->last_mode = STANDARD_KNN;
  if (si->replicas) return si->pick()->flat->topk(query, k, qp, order);
  if (scan_tuning().shard_exchange == 1 && k) return sharded_topk_rccl(si, query, k, qp, order);
  if (si->n() == 1) return si->shards[0]->flat->topk(query, k, qp, order);
  // fan out: one top-k task per shard; the shard workers batch the tasks of concurrent callers into shared passes
  std::vector<TopkJob> jobs(si->n(), TopkJob{query, k, qp ? qp->timeoutCtx : nullptr, BY_SCORE});
  Completion c;
  c.remaining = (int)si->n();
  for (size_t i = 0; i < si->n(); i++) {
    si->shards[i]->flat->last_mode = STANDARD_KNN;
    post(si->shards[i].get(), Task{nullptr, &jobs[i], &c});
  }
  wait_for(c);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<VecSimQueryReply *> replies(si->n(), nullptr);
  std::exception_ptr err;
  for (size_t i = 0; i < si->n(); i++) {
    replies[i] = jobs[i].reply;
    if (jobs[i].err && !err) err = jobs[i].err;
  }
  if (err) {
    for (VecSimQueryReply *r : replies) VecSimQueryReply_Free(r);
    std::rethrow_exception(err);
  }
  VecSimQueryReply *out = merge_replies(replies, k, true, order);
  si->merges++;
  si->merge_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return out;
}

VecSimQueryReply *sharded_range(RSGPU_ShardedIndex *si, const void *query, double radius, VecSimQueryParams *qp,
                                VecSimQueryReply_Order order) {
  DeviceGuard dg;
  si->last_mode = RANGE_QUERY;
  if (si->replicas) return si->pick()->flat->range(query, radius, qp, order);
  if (si->n() == 1) return si->shards[0]->flat->range(query, radius, qp, order);
  std::vector<VecSimQueryReply *> replies(si->n(), nullptr);
  std::vector<std::function<void()>> jobs;
  for (size_t i = 0; i < si->n(); i++) {
    FlatIndex *f = si->shards[i]->flat;
    VecSimQueryReply **slot = &replies[i];
    jobs.push_back([=] { *slot = f->range(query, radius, qp, BY_SCORE); });
  }
  try {
    run_on_shards(si, jobs);
  } catch (...) {
    for (VecSimQueryReply *r : replies) VecSimQueryReply_Free(r);
    throw;
  }
  return merge_replies(replies, 0, false, order);
}

// the reference's decision tree over the WHOLE index (vectors and labels summed over the shards)
bool sharded_prefer_adhoc(RSGPU_ShardedIndex *si, size_t subset, size_t k, bool initial_check) {
  (void)k;
  const bool res = FlatIndex::prefer_adhoc_rule(sharded_size(si), sharded_label_count(si), si->shards[0]->flat->dim, subset);
  si->last_mode = res ? (initial_check ? HYBRID_ADHOC_BF : HYBRID_BATCHES_TO_ADHOC_BF) : HYBRID_BATCHES;
  return res;
}

void sharded_reserve(RSGPU_ShardedIndex *si, size_t rows) {
  DeviceGuard dg;
  const size_t per = si->replicas ? rows : (rows + si->n() - 1) / si->n();
  for (auto &s : si->shards) s->flat->reserve(per);
}

long sharded_add_philox_rows(RSGPU_ShardedIndex *si, uint64_t seed, uint64_t first_index, size_t n, size_t first_label) {
  DeviceGuard dg;
  long added = 0;
  if (si->replicas) {
    for (auto &s : si->shards) added = s->flat->add_philox_rows(seed, first_index, n, first_label);
    return added;
  }
  const size_t per = (n + si->n() - 1) / si->n();
  for (size_t i = 0; i < si->n(); i++) {
    const size_t a = std::min(n, i * per), b = std::min(n, a + per);
    if (b > a) added += si->shards[i]->flat->add_philox_rows(seed, first_index + a, b - a, first_label + a);
  }
  return added;
}

// ---- batch iterator ---------------------------------------------------------------------------------------------------
// Every shard has its own iterator (keys of its rows, resumable select).  Next(n): every shard whose look-ahead buffer
// holds fewer than n results is asked for the difference -- all shards at once, the first call is the scan itself --
// then the n best of the buffers' fronts are taken by (score, label).  What stays in the buffers is the head start of
// the next call.  A label lives on one shard, so multi-value de-duplication stays inside the shard iterators.
struct ShardedBatchIterator {
  RSGPU_ShardedIndex *si = nullptr;
  std::vector<VecSimBatchIterator *> its;
  std::vector<std::deque<VecSimQueryResult>> buf;
};

ShardedBatchIterator *sharded_batch_new(RSGPU_ShardedIndex *si, const void *query, VecSimQueryParams *qp) {
  DeviceGuard dg;
  std::unique_ptr<ShardedBatchIterator> it(new ShardedBatchIterator());
  it->si = si;
  const size_t m = si->replicas ? 1 : si->n();
  const size_t first = si->replicas ? (size_t)(si->rr++ % si->n()) : 0;
  for (size_t i = 0; i < m; i++) {
    VecSimBatchIterator *b = VecSimBatchIterator_New(&si->handles[first + i], query, qp);
    if (!b) {
      for (VecSimBatchIterator *x : it->its) VecSimBatchIterator_Free(x);
      throw std::runtime_error(last_error());
    }
    it->its.push_back(b);
  }
  it->buf.resize(m);
  return it.release();
}

bool sharded_batch_has_next(ShardedBatchIterator *it) {
  for (size_t i = 0; i < it->its.size(); i++)
    if (!it->buf[i].empty() || VecSimBatchIterator_HasNext(it->its[i])) return true;
  return false;
}

VecSimQueryReply *sharded_batch_next(ShardedBatchIterator *it, size_t n, VecSimQueryReply_Order order) {
  DeviceGuard dg;
  RSGPU_ShardedIndex *si = it->si;
  si->last_mode = HYBRID_BATCHES;
  if (it->its.size() == 1) return VecSimBatchIterator_Next(it->its[0], n, order);
  std::vector<VecSimQueryReply *> replies(it->its.size(), nullptr);
  {
    std::vector<std::function<void()>> jobs;
    for (size_t i = 0; i < it->its.size(); i++) {
      const size_t have = it->buf[i].size();
      VecSimBatchIterator *b = it->its[i];
      VecSimQueryReply **slot = &replies[i];
      if (have >= n || !VecSimBatchIterator_HasNext(b)) {
        jobs.push_back(nullptr);
        continue;
      }
      const size_t need = n - have;
      jobs.push_back([=] {
        *slot = VecSimBatchIterator_Next(b, need, BY_SCORE);
        if (!*slot) throw std::runtime_error(last_error());
      });
    }
    try {
      run_on_shards(si, jobs);
    } catch (...) {
      for (VecSimQueryReply *r : replies) VecSimQueryReply_Free(r);
      throw;
    }
  }
  bool timed = false;
  for (size_t i = 0; i < replies.size(); i++) {
    VecSimQueryReply *r = replies[i];
    if (!r) continue;
    timed |= r->code == VecSim_QueryReply_TimedOut;
    for (size_t j = 0; j < r->len; j++) it->buf[i].push_back(r->results[j]);
    VecSimQueryReply_Free(r);
  }
  if (timed) return new_reply(0, VecSim_QueryReply_TimedOut);
  std::vector<VecSimQueryResult> out;
  while (out.size() < n) {
    int best = -1;
    for (size_t i = 0; i < it->buf.size(); i++)
      if (!it->buf[i].empty() && (best < 0 || by_score_then_id(it->buf[i].front(), it->buf[best].front()))) best = (int)i;
    if (best < 0) break;
    out.push_back(it->buf[best].front());
    it->buf[best].pop_front();
  }
  VecSimQueryReply *r = new_reply(out.size(), VecSim_QueryReply_OK);
  if (!out.empty()) memcpy(r->results, out.data(), out.size() * sizeof(VecSimQueryResult));
  if (order == BY_ID)
    std::sort(r->results, r->results + r->len, [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return a.id < b.id; });
  return r;
}

void sharded_batch_reset(ShardedBatchIterator *it) {
  DeviceGuard dg;
  for (size_t i = 0; i < it->its.size(); i++) {
    VecSimBatchIterator_Reset(it->its[i]);
    it->buf[i].clear();
  }
}

void sharded_batch_free(ShardedBatchIterator *it) {
  DeviceGuard dg;
  if (!it) return;
  for (VecSimBatchIterator *b : it->its) VecSimBatchIterator_Free(b);
  delete it;
}

// ---- ad-hoc context -----------------------------------------------------------------------------------------------------
struct ShardedAdhoc {
  RSGPU_ShardedIndex *si = nullptr;
  std::vector<VecSimAdhocBfCtx *> ctx;
  size_t first = 0;  // replica mode: the replica this context reads
};

ShardedAdhoc *sharded_adhoc_new(RSGPU_ShardedIndex *si, const void *query) {
  DeviceGuard dg;
  std::unique_ptr<ShardedAdhoc> a(new ShardedAdhoc());
  a->si = si;
  const size_t m = si->replicas ? 1 : si->n();
  a->first = si->replicas ? (size_t)(si->rr++ % si->n()) : 0;
  for (size_t i = 0; i < m; i++) {
    VecSimAdhocBfCtx *c = VecSimIndex_AdhocBfCtx_New(&si->handles[a->first + i], query);
    if (!c) {
      for (VecSimAdhocBfCtx *x : a->ctx) VecSimIndex_AdhocBfCtx_Free(x);
      throw std::runtime_error(last_error());
    }
    a->ctx.push_back(c);
  }
  return a.release();
}

void sharded_adhoc_distances(ShardedAdhoc *a, const size_t *labels, double *out, size_t count) {
  DeviceGuard dg;
  RSGPU_ShardedIndex *si = a->si;
  si->last_mode = HYBRID_ADHOC_BF;
  if (a->ctx.size() == 1) {
    VecSimIndex_AdhocBfCtx_GetExactDistances(a->ctx[0], labels, out, count);
    return;
  }
  const size_t m = a->ctx.size();
  std::vector<std::vector<size_t>> lab(m), at(m);
  for (size_t i = 0; i < count; i++) {
    out[i] = NAN;  // a label no shard holds
    for (size_t s = 0; s < m; s++)
      if (si->shards[s]->flat->contains(labels[i])) {
        lab[s].push_back(labels[i]);
        at[s].push_back(i);
        break;
      }
  }
  std::vector<std::vector<double>> d(m);
  std::vector<std::function<void()>> jobs;
  for (size_t s = 0; s < m; s++) {
    if (lab[s].empty()) {
      jobs.push_back(nullptr);
      continue;
    }
    d[s].resize(lab[s].size());
    VecSimAdhocBfCtx *c = a->ctx[s];
    const size_t *lp = lab[s].data();
    double *dp = d[s].data();
    const size_t cnt = lab[s].size();
    jobs.push_back([=] { VecSimIndex_AdhocBfCtx_GetExactDistances(c, lp, dp, cnt); });
  }
  run_on_shards(si, jobs);
  for (size_t s = 0; s < m; s++)
    for (size_t j = 0; j < at[s].size(); j++) out[at[s][j]] = d[s][j];
}

void sharded_adhoc_free(ShardedAdhoc *a) {
  DeviceGuard dg;
  if (!a) return;
  for (VecSimAdhocBfCtx *c : a->ctx) VecSimIndex_AdhocBfCtx_Free(c);
  delete a;
}

}  // namespace rsgpu

#define SH_TRY try {
#define SH_CATCH(si, where, ret)                                      \
  }                                                                   \
  catch (const std::exception &e) {                                   \
    last_error() = std::string(where) + ": " + e.what();              \
    logf((si) ? (si)->log_ctx : nullptr, "warning", "%s", last_error().c_str()); \
    return ret;                                                       \
  }                                                                   \
  catch (...) {                                                       \
    last_error() = std::string(where) + ": unknown error";            \
    return ret;                                                       \
  }

extern "C" {

RSGPU_ShardedIndex *RSGPU_ShardedIndex_New(const VecSimParams *params, int n_shards, const int *devices, int replicas) {
  if (!params || n_shards < 1 || n_shards > 64) return nullptr;
  try {
    if (params->algo != VecSimAlgo_BF) throw std::runtime_error("only VecSimAlgo_BF (FLAT) is served");
    return sharded_new(params->algoParams.bfParams, params->logCtx, n_shards, devices, replicas != 0);
  } catch (const std::exception &e) {
    last_error() = std::string("RSGPU_ShardedIndex_New: ") + e.what();
    logf(params->logCtx, "warning", "%s", last_error().c_str());
    return nullptr;
  }
}

void RSGPU_ShardedIndex_Free(RSGPU_ShardedIndex *si) { sharded_free(si); }

int RSGPU_ShardedIndex_NumShards(RSGPU_ShardedIndex *si) { return si ? (int)si->shards.size() : 0; }
int RSGPU_ShardedIndex_ShardDevice(RSGPU_ShardedIndex *si, int shard) {
  return (si && shard >= 0 && shard < (int)si->shards.size()) ? si->shards[shard]->device : -1;
}
VecSimIndex *RSGPU_ShardedIndex_Shard(RSGPU_ShardedIndex *si, int shard) {
  return (si && shard >= 0 && shard < (int)si->shards.size()) ? &si->handles[shard] : nullptr;
}
/* the ABI handle's shards, when the handle came from VecSimIndex_New under the "shards" knob (NULL otherwise) */
RSGPU_ShardedIndex *RSGPU_ShardedIndex_FromHandle(VecSimIndex *index) { return index ? index->sharded : nullptr; }

size_t RSGPU_ShardedIndex_IndexSize(RSGPU_ShardedIndex *si) { return si ? sharded_size(si) : 0; }

void RSGPU_ShardedIndex_GetExchangeStats(RSGPU_ShardedIndex *si, uint64_t out[2], int reset) {
  if (!si || !out) return;
  out[0] = si->merges.load();
  out[1] = si->merge_ns.load();
  if (reset) si->merges = si->merge_ns = 0;
}
/* the same for the queries that took the RCCL exchange (knob "shard_exchange" = 1): out[0] queries, out[1] nanoseconds of
 * the exchange (H2D of the winners + all-gather + merge kernel + sync), out[2] ranks of the communicator (0: not created yet) */
void RSGPU_ShardedIndex_GetRcclStats(RSGPU_ShardedIndex *si, uint64_t out[3], int reset) {
  if (!si || !out) return;
  out[0] = si->rccl_queries.load();
  out[1] = si->rccl_ns.load();
  out[2] = si->comms.size();
  if (reset) si->rccl_queries = si->rccl_ns = 0;
}

int RSGPU_ShardedIndex_AddVector(RSGPU_ShardedIndex *si, const void *blob, size_t label) {
  if (!si || !blob) return 0;
  SH_TRY
  return sharded_add(si, blob, label);
  SH_CATCH(si, "RSGPU_ShardedIndex_AddVector", 0)
}

int RSGPU_ShardedIndex_DeleteVector(RSGPU_ShardedIndex *si, size_t label) {
  if (!si) return 0;
  SH_TRY
  return sharded_remove(si, label);
  SH_CATCH(si, "RSGPU_ShardedIndex_DeleteVector", 0)
}

double RSGPU_ShardedIndex_GetDistanceFrom(RSGPU_ShardedIndex *si, size_t label, const void *normalized_blob) {
  if (!si || !normalized_blob) return NAN;
  SH_TRY
  return sharded_distance_from(si, label, normalized_blob);
  SH_CATCH(si, "RSGPU_ShardedIndex_GetDistanceFrom", NAN)
}

VecSimQueryReply *RSGPU_ShardedIndex_TopKQuery(RSGPU_ShardedIndex *si, const void *query, size_t k,
                                               VecSimQueryParams *qp, VecSimQueryReply_Order order) {
  if (!si || !query) return nullptr;
  SH_TRY
  return sharded_topk(si, query, k, qp, order);
  SH_CATCH(si, "RSGPU_ShardedIndex_TopKQuery", nullptr)
}

VecSimQueryReply *RSGPU_ShardedIndex_RangeQuery(RSGPU_ShardedIndex *si, const void *query, double radius,
                                                VecSimQueryParams *qp, VecSimQueryReply_Order order) {
  if (!si || !query) return nullptr;
  SH_TRY
  return sharded_range(si, query, radius, qp, order);
  SH_CATCH(si, "RSGPU_ShardedIndex_RangeQuery", nullptr)
}

}  // extern "C"
