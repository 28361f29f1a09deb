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
#include <mutex>
#include <thread>

#include "flat_index.hpp"
#include "rsgpu_ext.h"

using namespace rsgpu;

namespace {

struct Job {
  enum Kind { NONE, TOPK, RANGE, STOP } kind = NONE;
  const void *query = nullptr;
  size_t k = 0;
  double radius = 0;
  VecSimQueryParams *qp = nullptr;
};

struct Shard {
  FlatIndex *flat = nullptr;
  int device = 0;
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  Job job;
  uint64_t posted = 0, done = 0;  // generation counters
  VecSimQueryReply *reply = nullptr;
  std::string error;
};

}  // namespace

struct RSGPU_ShardedIndex {
  std::vector<std::unique_ptr<Shard>> shards;
  bool replicas = false;
  bool multi = false;
  void *log_ctx = nullptr;
  std::mutex done_mu;
  std::condition_variable done_cv;
  std::mutex query_mu;  // one fan-out at a time (the workers hold one job slot each)
  std::atomic<uint64_t> rr{0};
  std::vector<VecSimIndex> handles;  // borrowed views for RSGPU_ShardedIndex_Shard
};

static void worker_main(RSGPU_ShardedIndex *si, Shard *s) {
  (void)hipSetDevice(s->device);
  uint64_t seen = 0;
  for (;;) {
    Job job;
    {
      std::unique_lock<std::mutex> g(s->mu);
      s->cv.wait(g, [&] { return s->posted != seen; });
      seen = s->posted;
      job = s->job;
    }
    if (job.kind == Job::STOP) return;
    VecSimQueryReply *r = nullptr;
    std::string err;
    try {
      if (job.kind == Job::TOPK) r = s->flat->topk(job.query, job.k, job.qp, BY_SCORE);
      else if (job.kind == Job::RANGE) r = s->flat->range(job.query, job.radius, job.qp, BY_SCORE);
    } catch (const std::exception &e) {
      err = e.what();
    } catch (...) {
      err = "unknown error";
    }
    {
      std::lock_guard<std::mutex> g(si->done_mu);
      s->reply = r;
      s->error = err;
      s->done = seen;
    }
    si->done_cv.notify_all();
  }
}

static void post_all(RSGPU_ShardedIndex *si, const Job &job) {
  for (auto &s : si->shards) {
    {
      std::lock_guard<std::mutex> g(s->mu);
      s->job = job;
      s->posted++;
    }
    s->cv.notify_one();
  }
  std::unique_lock<std::mutex> g(si->done_mu);
  si->done_cv.wait(g, [&] {
    for (auto &s : si->shards)
      if (s->done != s->posted) return false;
    return true;
  });
}

// K-way merge of per-shard replies (each ascending by (score, id)); frees them.
static VecSimQueryReply *merge_replies(RSGPU_ShardedIndex *si, size_t k, bool cut, VecSimQueryReply_Order order) {
  size_t total = 0;
  bool timed_out_any = false;
  std::string err;
  for (auto &s : si->shards) {
    if (!s->error.empty()) err = s->error;
    if (s->reply) {
      total += s->reply->len;
      timed_out_any |= s->reply->code == VecSim_QueryReply_TimedOut;
    }
  }
  VecSimQueryReply *out = nullptr;
  if (err.empty()) {
    if (timed_out_any) {
      out = new_reply(0, VecSim_QueryReply_TimedOut);
    } else {
      const size_t take = cut ? std::min(k, total) : total;
      out = new_reply(take, VecSim_QueryReply_OK);
      std::vector<size_t> pos(si->shards.size(), 0);
      for (size_t n = 0; n < take; n++) {
        int best = -1;
        for (size_t i = 0; i < si->shards.size(); i++) {
          VecSimQueryReply *r = si->shards[i]->reply;
          if (!r || pos[i] >= r->len) continue;
          if (best < 0) { best = (int)i; continue; }
          const VecSimQueryResult &a = r->results[pos[i]], &b = si->shards[best]->reply->results[pos[best]];
          if (a.score != b.score ? a.score < b.score : a.id < b.id) best = (int)i;
        }
        out->results[n] = si->shards[best]->reply->results[pos[best]++];
      }
      if (order == BY_ID)
        std::sort(out->results, out->results + out->len,
                  [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return a.id < b.id; });
    }
  }
  for (auto &s : si->shards) {
    VecSimQueryReply_Free(s->reply);
    s->reply = nullptr;
    s->error.clear();
  }
  if (!err.empty()) throw std::runtime_error(err);
  return out;
}

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
  RSGPU_ShardedIndex *si = nullptr;
  SH_TRY
  if (params->algo != VecSimAlgo_BF) throw std::runtime_error("only VecSimAlgo_BF (FLAT) is served");
  std::string why;
  if (!device_available(&why)) throw std::runtime_error(why);
  int ndev = 0;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  int prev = 0;
  HIP_CHECK(hipGetDevice(&prev));
  si = new RSGPU_ShardedIndex();
  si->replicas = replicas != 0;
  si->multi = params->algoParams.bfParams.multi;
  si->log_ctx = params->logCtx;
  for (int i = 0; i < n_shards; i++) {
    const int dev = devices ? devices[i] : i % ndev;
    if (dev < 0 || dev >= ndev) throw std::runtime_error("RSGPU_ShardedIndex_New: no such device");
    HIP_CHECK(hipSetDevice(dev));
    std::unique_ptr<Shard> s(new Shard());
    s->device = dev;
    s->flat = new FlatIndex(params->algoParams.bfParams, params->logCtx);
    si->shards.push_back(std::move(s));
  }
  HIP_CHECK(hipSetDevice(prev));
  si->handles.resize(si->shards.size());
  for (size_t i = 0; i < si->shards.size(); i++) {
    si->handles[i].flat = si->shards[i]->flat;
    if (!si->replicas) si->shards[i]->worker = std::thread(worker_main, si, si->shards[i].get());
  }
  return si;
  }
  catch (const std::exception &e) {
    last_error() = std::string("RSGPU_ShardedIndex_New: ") + e.what();
    logf(params->logCtx, "warning", "%s", last_error().c_str());
    if (si) {
      for (auto &s : si->shards) delete s->flat;
      delete si;
    }
    return nullptr;
  }
}

void RSGPU_ShardedIndex_Free(RSGPU_ShardedIndex *si) {
  if (!si) return;
  for (auto &s : si->shards) {
    if (s->worker.joinable()) {
      {
        std::lock_guard<std::mutex> g(s->mu);
        s->job.kind = Job::STOP;
        s->posted++;
      }
      s->cv.notify_one();
      s->worker.join();
    }
    try {
      delete s->flat;
    } catch (...) {
    }
  }
  delete si;
}

int RSGPU_ShardedIndex_NumShards(RSGPU_ShardedIndex *si) { return si ? (int)si->shards.size() : 0; }
int RSGPU_ShardedIndex_ShardDevice(RSGPU_ShardedIndex *si, int shard) {
  return (si && shard >= 0 && shard < (int)si->shards.size()) ? si->shards[shard]->device : -1;
}
VecSimIndex *RSGPU_ShardedIndex_Shard(RSGPU_ShardedIndex *si, int shard) {
  return (si && shard >= 0 && shard < (int)si->shards.size()) ? &si->handles[shard] : nullptr;
}

size_t RSGPU_ShardedIndex_IndexSize(RSGPU_ShardedIndex *si) {
  if (!si) return 0;
  if (si->replicas) return si->shards[0]->flat->size();
  size_t n = 0;
  for (auto &s : si->shards) n += s->flat->size();
  return n;
}

// A label lives on exactly one shard: an existing label is overwritten (single-value) or extended (multi-value) where
// it is, a new one goes to the emptiest shard.  Replicas: every shard gets the vector.
int RSGPU_ShardedIndex_AddVector(RSGPU_ShardedIndex *si, const void *blob, size_t label) {
  if (!si || !blob) return 0;
  SH_TRY
  if (si->replicas) {
    int r = 0;
    for (auto &s : si->shards) r = s->flat->add(blob, label);
    return r;
  }
  Shard *target = nullptr;
  for (auto &s : si->shards)
    if (s->flat->contains(label)) { target = s.get(); break; }
  if (!target) {
    size_t best = SIZE_MAX;
    for (auto &s : si->shards) {
      const size_t n = s->flat->size();
      if (n < best) { best = n; target = s.get(); }
    }
  }
  return target->flat->add(blob, label);
  SH_CATCH(si, "RSGPU_ShardedIndex_AddVector", 0)
}

int RSGPU_ShardedIndex_DeleteVector(RSGPU_ShardedIndex *si, size_t label) {
  if (!si) return 0;
  SH_TRY
  int removed = 0;
  for (auto &s : si->shards) {
    const int r = s->flat->remove(label);
    removed = si->replicas ? r : removed + r;
  }
  return removed;
  SH_CATCH(si, "RSGPU_ShardedIndex_DeleteVector", 0)
}

double RSGPU_ShardedIndex_GetDistanceFrom(RSGPU_ShardedIndex *si, size_t label, const void *normalized_blob) {
  if (!si || !normalized_blob) return NAN;
  SH_TRY
  for (auto &s : si->shards) {
    if (!s->flat->contains(label)) continue;
    return s->flat->distance_from(label, normalized_blob);
  }
  return NAN;
  SH_CATCH(si, "RSGPU_ShardedIndex_GetDistanceFrom", NAN)
}

VecSimQueryReply *RSGPU_ShardedIndex_TopKQuery(RSGPU_ShardedIndex *si, const void *query, size_t k,
                                               VecSimQueryParams *qp, VecSimQueryReply_Order order) {
  if (!si || !query) return nullptr;
  SH_TRY
  if (si->replicas) {
    Shard *s = si->shards[si->rr++ % si->shards.size()].get();
    return s->flat->topk(query, k, qp, order);
  }
  if (si->shards.size() == 1) return si->shards[0]->flat->topk(query, k, qp, order);
  std::lock_guard<std::mutex> q(si->query_mu);
  Job job;
  job.kind = Job::TOPK;
  job.query = query;
  job.k = k;
  job.qp = qp;
  post_all(si, job);
  return merge_replies(si, k, true, order);
  SH_CATCH(si, "RSGPU_ShardedIndex_TopKQuery", nullptr)
}

VecSimQueryReply *RSGPU_ShardedIndex_RangeQuery(RSGPU_ShardedIndex *si, const void *query, double radius,
                                                VecSimQueryParams *qp, VecSimQueryReply_Order order) {
  if (!si || !query) return nullptr;
  SH_TRY
  if (si->replicas) {
    Shard *s = si->shards[si->rr++ % si->shards.size()].get();
    return s->flat->range(query, radius, qp, order);
  }
  if (si->shards.size() == 1) return si->shards[0]->flat->range(query, radius, qp, order);
  std::lock_guard<std::mutex> q(si->query_mu);
  Job job;
  job.kind = Job::RANGE;
  job.query = query;
  job.radius = radius;
  job.qp = qp;
  post_all(si, job);
  return merge_replies(si, 0, false, order);
  SH_CATCH(si, "RSGPU_ShardedIndex_RangeQuery", nullptr)
}

}  // extern "C"
