/* Prints the layout facts an iterator implemented outside the module depends on.  Compiled twice by
 * tests/test_iterator_abi.py: against the reference's own headers (-DPROBE_REFERENCE, where /root/reference exists) and
 * against include/rs_iterator.h; the two outputs must be identical. */
#include <stdio.h>
#include <stddef.h>
#include <string.h>
#ifdef PROBE_REFERENCE
#include "redisearch.h"
#include "iterator_api.h"
#include "types_ffi.h"
#include "index_result_rs.h"
#define TERM_OFFSETS term.borrowed.offsets
#else
#include "rs_iterator.h"
#define TERM_OFFSETS term.offsets
#endif
#define P(x) printf(#x " %zu\n", (size_t)(x))
int main(void) {
  P(sizeof(QueryIterator));
  P(offsetof(QueryIterator, type));
  P(offsetof(QueryIterator, atEOF));
  P(offsetof(QueryIterator, lastDocId));
  P(offsetof(QueryIterator, current));
  P(offsetof(QueryIterator, NumEstimated));
  P(offsetof(QueryIterator, Read));
  P(offsetof(QueryIterator, SkipTo));
  P(offsetof(QueryIterator, Revalidate));
  P(offsetof(QueryIterator, Free));
  P(offsetof(QueryIterator, Rewind));
  P(offsetof(QueryIterator, ProfileChildren));
  P(offsetof(QueryIterator, PrintProfile));
  P(sizeof(enum IteratorType));
  printf("status %d %d %d %d\n", ITERATOR_OK, ITERATOR_NOTFOUND, ITERATOR_EOF, ITERATOR_TIMEOUT);
  printf("validate %d %d %d %d\n", VALIDATE_OK, VALIDATE_MOVED, VALIDATE_ABORTED, VALIDATE_TIMEOUT);
  printf("types %d %d %d %d %d %d\n", (int)IteratorType_Union, (int)IteratorType_Intersect, (int)IteratorType_Not,
         (int)IteratorType_Empty, (int)IteratorType_IdListSorted, (int)IteratorType_Max);
  P(sizeof(t_docId));
  P(sizeof(t_fieldMask));
  t_fieldMask all = RS_FIELDMASK_ALL;
  printf("fieldmask_all %016llx%016llx\n", (unsigned long long)(all >> 64), (unsigned long long)all);
  /* where a Term record keeps its offsets slice (what RSOffsetVector_SetData is handed) */
  RSIndexResult r;
  memset(&r, 0, sizeof r);
  printf("term.offsets %zu\n", (size_t)((char *)&r.data.TERM_OFFSETS - (char *)&r));
  P(sizeof(r.data.TERM_OFFSETS));
  return 0;
}
