/* N caller threads on ONE index through the plain VecSim C ABI -- the shape of RediSearch's worker pool
 * (reference src/util/workers.c:58,104: WORKERS n threads, each running its own query's
 * VecSimIndex_TopKQuery -> reply iteration -> VecSimQueryReply_Free, src/iterators/hybrid_reader.c:374, :61-85, :543).
 *
 * Built as a small shared object so that bench.py / the tests can drive it with ctypes without Python threads in the
 * measurement:
 *   gcc -O2 -shared -fPIC -Iinclude examples/concurrent_callers.c -Lredisearch_amd/lib -lVectorSimilarity \
 *       -Wl,-rpath,$PWD/redisearch_amd/lib -lpthread -o libconcurrent_callers.so
 *
 * rs_callers_run: `threads` threads, thread t issues queries t, t + threads, ... (mod nq) back to back for `seconds`;
 * every call's latency goes to lat_ns[t * lat_cap + j]; the LAST reply seen for query i is kept in ids_out / scores_out
 * [i][k] so that the caller can hold coalesced answers to serial ones.  Returns the number of queries answered, or -1. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "VecSim/query_results.h"
#include "VecSim/vec_sim.h"

typedef struct {
  VecSimIndex *index;
  const uint8_t *queries;
  size_t query_bytes, nq, k;
  int t, threads;
  double seconds;
  uint64_t *lat_ns;
  size_t lat_cap, done;
  size_t *ids_out;
  double *scores_out;
  pthread_barrier_t *start;
  int failed;
} Caller;

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void *caller_main(void *arg) {
  Caller *c = (Caller *)arg;
  pthread_barrier_wait(c->start);
  const uint64_t t_end = now_ns() + (uint64_t)(c->seconds * 1e9);
  size_t i = (size_t)c->t;
  while (c->done < c->lat_cap) {
    const size_t qi = i % c->nq;
    const uint64_t t0 = now_ns();
    if (t0 >= t_end && c->done > 0) break;
    VecSimQueryReply *r = VecSimIndex_TopKQuery(c->index, c->queries + qi * c->query_bytes, c->k, NULL, BY_SCORE);
    if (!r || VecSimQueryReply_GetCode(r) != VecSim_QueryReply_OK) {
      c->failed = 1;
      VecSimQueryReply_Free(r);
      break;
    }
    /* what the reference does with a reply: walk it (hybrid_reader.c:61-85) */
    VecSimQueryReply_Iterator *it = VecSimQueryReply_GetIterator(r);
    size_t j = 0;
    while (VecSimQueryReply_IteratorHasNext(it)) {
      VecSimQueryResult *res = VecSimQueryReply_IteratorNext(it);
      if (j < c->k && c->ids_out) {
        c->ids_out[qi * c->k + j] = VecSimQueryResult_GetId(res);
        c->scores_out[qi * c->k + j] = VecSimQueryResult_GetScore(res);
      }
      j++;
    }
    VecSimQueryReply_IteratorFree(it);
    VecSimQueryReply_Free(r);
    c->lat_ns[(size_t)c->t * c->lat_cap + c->done] = now_ns() - t0;
    c->done++;
    i += (size_t)c->threads;
  }
  return NULL;
}

long rs_callers_run(VecSimIndex *index, const void *queries, size_t query_bytes, size_t nq, size_t k, int threads,
                    double seconds, uint64_t *lat_ns, size_t lat_cap, size_t *counts, size_t *ids_out, double *scores_out,
                    double *elapsed_s) {
  if (!index || !queries || !nq || threads < 1 || threads > 256 || !lat_ns || !lat_cap) return -1;
  pthread_t th[256];
  Caller cs[256];
  pthread_barrier_t start;
  pthread_barrier_init(&start, NULL, (unsigned)threads + 1);
  memset(cs, 0, sizeof cs);
  for (int t = 0; t < threads; t++) {
    cs[t] = (Caller){index, (const uint8_t *)queries, query_bytes, nq, k, t, threads, seconds, lat_ns, lat_cap, 0,
                     ids_out, scores_out, &start, 0};
    if (pthread_create(&th[t], NULL, caller_main, &cs[t]) != 0) return -1;
  }
  pthread_barrier_wait(&start);
  const uint64_t t0 = now_ns();
  long total = 0;
  int failed = 0;
  for (int t = 0; t < threads; t++) {
    pthread_join(th[t], NULL);
    total += (long)cs[t].done;
    failed |= cs[t].failed;
    if (counts) counts[t] = cs[t].done;
  }
  if (elapsed_s) *elapsed_s = (double)(now_ns() - t0) / 1e9;
  pthread_barrier_destroy(&start);
  return failed ? -1 : total;
}
