/* n caller threads through the plain C ABI, each issuing RSGPU_HybridQuery calls back to back on prepared argument blocks -- the shape
 * of a RediSearch deployment with WORKERS n (reference src/util/workers.c:58,104: queries run on a thread pool; each worker owns its
 * query's iterators and buffers, the index and the posting lists are shared).  Thread t cycles over the blocks t, t + n, t + 2n, ...
 * (every block has its own output arrays).  Built on the spot by bench.py / the tests:
 *   gcc -O2 -shared -fPIC -Iinclude examples/concurrent_hybrid_callers.c -Lredisearch_amd/lib -lVectorSimilarity -lpthread */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#include "rsgpu_search.h"

typedef struct {
  RSGPU_HybridQueryArgs **blocks;
  size_t n_blocks;
  int t, threads;
  double seconds;
  uint64_t *lat_ns;
  size_t lat_cap;
  uint64_t count;
  int failed;
  pthread_barrier_t *go;
} worker_t;

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void *work(void *p) {
  worker_t *w = (worker_t *)p;
  pthread_barrier_wait(w->go);
  const uint64_t stop = now_ns() + (uint64_t)(w->seconds * 1e9);
  size_t i = (size_t)w->t;
  for (;;) {
    const uint64_t t0 = now_ns();
    if (t0 >= stop) break;
    if (RSGPU_HybridQuery(w->blocks[i]) != 0) {
      w->failed = 1;
      break;
    }
    if (w->count < w->lat_cap) w->lat_ns[w->count] = now_ns() - t0;
    w->count++;
    i += (size_t)w->threads;
    if (i >= w->n_blocks) i = (size_t)w->t % w->n_blocks;
  }
  return NULL;
}

/* returns the number of queries answered (-1: a call failed); lat_ns[threads][lat_cap], counts[threads] */
long rs_hybrid_callers_run(RSGPU_HybridQueryArgs **blocks, size_t n_blocks, int threads, double seconds, uint64_t *lat_ns,
                           size_t lat_cap, uint64_t *counts, double *elapsed_s) {
  if (threads < 1 || threads > 256 || n_blocks < (size_t)threads) return -1;
  pthread_t th[256];
  worker_t w[256];
  pthread_barrier_t go;
  pthread_barrier_init(&go, NULL, (unsigned)threads + 1);
  for (int t = 0; t < threads; t++) {
    w[t] = (worker_t){blocks, n_blocks, t, threads, seconds, lat_ns + (size_t)t * lat_cap, lat_cap, 0, 0, &go};
    pthread_create(&th[t], NULL, work, &w[t]);
  }
  pthread_barrier_wait(&go);
  const uint64_t t0 = now_ns();
  long total = 0;
  int failed = 0;
  for (int t = 0; t < threads; t++) {
    pthread_join(th[t], NULL);
    counts[t] = w[t].count;
    total += (long)w[t].count;
    failed |= w[t].failed;
  }
  *elapsed_s = (double)(now_ns() - t0) / 1e9;
  pthread_barrier_destroy(&go);
  return failed ? -1 : total;
}
