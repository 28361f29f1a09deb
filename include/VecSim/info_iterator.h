/*
 * VecSim/info_iterator.h -- the FT.DEBUG VECSIM_INFO field iterator of the VecSim C ABI.
 *
 * Replaces deps/VectorSimilarity/src/VecSim/info_iterator.h (absent submodule); included directly by
 * reference src/debug_commands.c:49 and, as upstream does, by VecSim/vec_sim.h.  The consumer is
 * VecSim_Reply_Info_Iterator (reference src/debug_commands.c:1664-1690): it sizes the reply with
 * NumberOfFields, then walks HasNextField/NextField and switches on fieldType, recursing into
 * iteratorValue for INFOFIELD_ITERATOR.  Freeing the top iterator frees its nested children (:1719).
 */
#ifndef VECSIM_INFO_ITERATOR_H
#define VECSIM_INFO_ITERATOR_H

#include "vec_sim_common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/debug_commands.c:1668-1686 switches over exactly these five. */
typedef enum {
  INFOFIELD_STRING,
  INFOFIELD_INT64,
  INFOFIELD_UINT64,
  INFOFIELD_FLOAT64,
  INFOFIELD_ITERATOR
} VecSim_InfoFieldType;

typedef union {
  double floatingPointValue;
  int64_t integerValue;
  uint64_t uintegerValue;
  const char *stringValue;
  VecSimDebugInfoIterator *iteratorValue;
} FieldValue;

typedef struct {
  const char *fieldName;
  VecSim_InfoFieldType fieldType;
  FieldValue fieldValue;
} VecSim_InfoField;

/* reference src/debug_commands.c:1665 */
size_t VecSimDebugInfoIterator_NumberOfFields(VecSimDebugInfoIterator *infoIterator);
/* reference src/debug_commands.c:1666 */
bool VecSimDebugInfoIterator_HasNextField(VecSimDebugInfoIterator *infoIterator);
/* reference src/debug_commands.c:1667 -- the field is borrowed from the iterator */
VecSim_InfoField *VecSimDebugInfoIterator_NextField(VecSimDebugInfoIterator *infoIterator);
/* reference src/debug_commands.c:1719 -- NULL-safe; frees nested iterators */
void VecSimDebugInfoIterator_Free(VecSimDebugInfoIterator *infoIterator);

#ifdef __cplusplus
}
#endif
#endif /* VECSIM_INFO_ITERATOR_H */
