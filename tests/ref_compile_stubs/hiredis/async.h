/* Test-only stand-in for the un-vendored deps/hiredis submodule: just enough declarations for the reference's
 * coordinator HEADERS (src/coord/rmr/conn.h, cluster.h) to parse while its VecSim callers are syntax-checked. */
#ifndef STUB_HIREDIS_ASYNC_H
#define STUB_HIREDIS_ASYNC_H
struct redisAsyncContext;
typedef struct redisAsyncContext redisAsyncContext;
typedef void(redisCallbackFn)(struct redisAsyncContext *, void *, void *);
#endif
