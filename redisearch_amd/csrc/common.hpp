// common.hpp -- shared plumbing of the MI355X FLAT engine (host side).
//
// HIP error handling, the process-wide hooks installed through the VecSim C ABI
// (reference src/module-init/module-init.c:147-151: memory functions, timeout callback, log
// callback), and small helpers.  gfx950 only; there is no CPU fallback anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>

#include "VecSim/vec_sim_common.h"

namespace rsgpu {

struct Hooks {
  VecSimMemoryFunctions mem{malloc, calloc, realloc, free};
  timeoutCallbackFunction timeout = nullptr;  // NULL => never times out
  logCallbackFunction log = nullptr;
  size_t thread_pool_size = 0;
};
Hooks &hooks();

void logf(void *ctx, const char *level, const char *fmt, ...);

struct HipError : std::runtime_error {
  hipError_t code;
  HipError(hipError_t c, const char *what, const char *file, int line)
      : std::runtime_error(std::string(what) + ": " + hipGetErrorString(c) + " (" + file + ":" + std::to_string(line) + ")"),
        code(c) {}
};

#define HIP_CHECK(expr)                                                \
  do {                                                                 \
    hipError_t _e = (expr);                                            \
    if (_e != hipSuccess) throw ::rsgpu::HipError(_e, #expr, __FILE__, __LINE__); \
  } while (0)

// True when a gfx950-class device is usable. Failing loudly is the contract: callers turn a false
// into NULL + a logged error, never into a CPU path.
bool device_available(std::string *why = nullptr);

inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// `a` before `b` in score order -- a STRICT WEAK order even when a score is NaN (a NaN element stored in the index, a NaN
// query): NaN sorts after every number and NaNs tie among themselves.  `a != b ? a < b : tie` is not one with NaNs around,
// and std::sort / std::partial_sort may then walk out of the range they were given.
inline bool score_before(double a, double b) {
  const bool an = a != a, bn = b != b;
  if (an || bn) return !an && bn;
  return a < b;
}
// (score, id) ascending with that order
template <typename S, typename I>
inline bool score_id_before(S sa, I ia, S sb, I ib) {
  if (score_before((double)sa, (double)sb)) return true;
  if (score_before((double)sb, (double)sa)) return false;
  return ia < ib;
}

inline size_t type_size(VecSimType t) {
  switch (t) {
    case VecSimType_FLOAT32: return 4;
    case VecSimType_FLOAT64: return 8;
    case VecSimType_BFLOAT16:
    case VecSimType_FLOAT16: return 2;
    case VecSimType_INT8:
    case VecSimType_UINT8: return 1;
    case VecSimType_INT32: return 4;
    case VecSimType_INT64: return 8;
  }
  return 0;
}

// Host allocations made on behalf of the caller go through the installed memory functions so that
// FT.INFO memory accounting stays sane (SURVEY.md 7.3 "drop-in honesty").
template <typename T>
T *host_alloc(size_t n) {
  return static_cast<T *>(hooks().mem.allocFunction(n * sizeof(T)));
}
inline void host_free(void *p) {
  if (p) hooks().mem.freeFunction(p);
}

// std allocator over the installed VecSimMemoryFunctions, counting live bytes into a caller-owned counter: the label
// maps of an index (hundreds of MB at 10 M labels) are allocated by the module's allocator and show up in FT.INFO
// memory accounting (SURVEY.md 7.3 "drop-in honesty").
template <typename T>
struct HookAlloc {
  using value_type = T;
  size_t *live = nullptr;  // bytes currently allocated through this allocator family (guarded by the index lock)
  HookAlloc() = default;
  explicit HookAlloc(size_t *counter) : live(counter) {}
  template <typename U>
  HookAlloc(const HookAlloc<U> &o) : live(o.live) {}
  T *allocate(size_t n) {
    void *p = hooks().mem.allocFunction(n * sizeof(T));
    if (!p) throw std::bad_alloc();
    if (live) *live += n * sizeof(T);
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t n) {
    hooks().mem.freeFunction(p);
    if (live) *live -= n * sizeof(T);
  }
  template <typename U>
  bool operator==(const HookAlloc<U> &o) const { return live == o.live; }
  template <typename U>
  bool operator!=(const HookAlloc<U> &o) const { return live != o.live; }
};

}  // namespace rsgpu
