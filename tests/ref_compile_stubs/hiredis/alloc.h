/* Test-only stand-in for the un-vendored deps/hiredis submodule (see async.h). */
#ifndef STUB_HIREDIS_ALLOC_H
#define STUB_HIREDIS_ALLOC_H
#include <stddef.h>
typedef struct hiredisAllocFuncs {
  void *(*mallocFn)(size_t);
  void *(*callocFn)(size_t, size_t);
  void *(*reallocFn)(void *, size_t);
  char *(*strdupFn)(const char *);
  void (*freeFn)(void *);
} hiredisAllocFuncs;
hiredisAllocFuncs hiredisSetAllocators(hiredisAllocFuncs *ha);
#endif
