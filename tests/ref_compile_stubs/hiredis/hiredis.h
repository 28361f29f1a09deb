/* Test-only stand-in for the un-vendored deps/hiredis submodule (see async.h). */
#ifndef STUB_HIREDIS_H
#define STUB_HIREDIS_H
#include <stddef.h>
#include "async.h"
#include "alloc.h"
#define REDIS_OK 0
#define REDIS_ERR -1
typedef struct redisReply redisReply;
#endif
