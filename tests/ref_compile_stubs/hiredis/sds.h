/* Test-only stand-in for the un-vendored deps/hiredis submodule: the sds prototypes the reference headers mention. */
#ifndef STUB_SDS_H
#define STUB_SDS_H
#include <stddef.h>
#include <stdarg.h>
#include <sys/types.h>
typedef char *sds;
sds sdsnew(const char *); sds sdsempty(void); sds sdsnewlen(const void*, size_t); size_t sdslen(const sds); void sdsfree(sds);
sds sdscat(sds, const char*); sds sdscatlen(sds, const void*, size_t); sds sdscatprintf(sds, const char*, ...); sds sdscatfmt(sds, const char*, ...);
sds sdsdup(const sds); sds sdscatsds(sds, const sds); void sdsclear(sds); sds sdstrim(sds, const char*); sds sdscpy(sds, const char*); sds sdscpylen(sds,const char*,size_t);
int sdscmp(const sds, const sds); sds sdsjoin(char **, int, char *); void sdstolower(sds); void sdstoupper(sds); sds sdsfromlonglong(long long);
sds *sdssplitlen(const char*, int, const char*, int, int*); void sdsfreesplitres(sds*, int); sds sdscatrepr(sds, const char*, size_t); sds *sdssplitargs(const char*, int*);
sds sdsgrowzero(sds, size_t); sds sdsMakeRoomFor(sds, size_t); void sdsIncrLen(sds, ssize_t); void sdsrange(sds, ssize_t, ssize_t);
#endif
