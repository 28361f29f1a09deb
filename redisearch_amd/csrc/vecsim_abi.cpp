// vecsim_abi.cpp -- the VecSim C ABI (include/VecSim/*.h) implemented on the MI355X FLAT engine,
// plus the non-ABI extensions of include/rsgpu_ext.h.
//
// This translation unit is the drop-in seam: every function below is bound by the reference at the
// file:line quoted in the headers.  No exception crosses the boundary: failures become NULL / NaN /
// an error code plus a message through the installed log callback (SURVEY.md 8b "Errors").
#include <algorithm>
#include <cmath>
#include <memory>
#include <strings.h>

#include "flat_index.hpp"
#include "sharded_index.hpp"
#include "VecSim/vec_sim_debug.h"
#include "rsgpu_ext.h"

using namespace rsgpu;

namespace rsgpu {
void normalize_blob(void *blob, size_t dim, VecSimType type);
}

struct VecSimBatchIterator {
  BatchIterator it;
  ShardedBatchIterator *sh = nullptr;  // the iterator of a sharded handle (then `it` is unused)
};
struct VecSimAdhocBfCtx {
  AdhocCtx a;
  ShardedAdhoc *sh = nullptr;
};
static void *log_ctx_of(VecSimIndex *index) { return index->sharded ? sharded_log_ctx(index->sharded) : index->flat->log_ctx; }
struct VecSimDebugInfoIterator {
  std::vector<VecSim_InfoField> fields;
  std::vector<std::string> strings;
  size_t pos = 0;
};

static void set_error(void *log_ctx, const char *where, const char *what) {
  last_error() = std::string(where) + ": " + what;
  logf(log_ctx, VecSimCommonStrings_LOG_WARNING_STRING, "%s", last_error().c_str());
}
#define ABI_TRY try {
#define ABI_CATCH(lctx, where, failval)              \
  }                                                  \
  catch (const std::exception &e) {                  \
    set_error(lctx, where, e.what());                \
    return failval;                                  \
  }                                                  \
  catch (...) {                                      \
    set_error(lctx, where, "unknown error");         \
    return failval;                                  \
  }

static bool type_supported(VecSimType t) {
  return t == VecSimType_FLOAT32 || t == VecSimType_FLOAT64 || t == VecSimType_FLOAT16 || t == VecSimType_BFLOAT16 ||
         t == VecSimType_INT8 || t == VecSimType_UINT8;
}

extern "C" {

// ---- lifecycle -----------------------------------------------------------------------------------
VecSimIndex *VecSimIndex_New(const VecSimParams *params) {
  if (!params) return nullptr;
  void *lctx = params->logCtx;
  ABI_TRY
  if (params->algo != VecSimAlgo_BF) {
    set_error(lctx, "VecSimIndex_New", "only VecSimAlgo_BF (FLAT) is served by the MI355X engine");
    return nullptr;
  }
  const BFParams &bf = params->algoParams.bfParams;
  if (bf.dim == 0 || !type_supported(bf.type) || (int)bf.metric < 0 || (int)bf.metric > (int)VecSimMetric_Cosine) {
    set_error(lctx, "VecSimIndex_New", "unsupported FLAT parameters (dim>0; type FLOAT32/FLOAT64/FLOAT16/BFLOAT16/INT8/UINT8)");
    return nullptr;
  }
  if ((bf.type == VecSimType_INT8 || bf.type == VecSimType_UINT8) && bf.dim > 65536) {
    set_error(lctx, "VecSimIndex_New", "INT8/UINT8 FLAT: dim <= 65536 (exact 32-bit integer sums)");
    return nullptr;
  }
  std::string why;
  if (!device_available(&why)) {  // no CPU fallback: fail loudly
    set_error(lctx, "VecSimIndex_New", why.c_str());
    return nullptr;
  }
  // "shards" knob: the same handle over several device shards -- every entry point below dispatches on it
  if (scan_tuning().shards > 1)
    return new VecSimIndex{nullptr, sharded_new(bf, lctx, scan_tuning().shards, nullptr, scan_tuning().shard_replicas != 0)};
  VecSimIndex *idx = new VecSimIndex{new FlatIndex(bf, lctx)};
  return idx;
  ABI_CATCH(lctx, "VecSimIndex_New", nullptr)
}

VecSimIndex *VecSimIndex_NewDisk(const VecSimParamsDisk *) { return nullptr; }

void VecSimIndex_Free(VecSimIndex *index) {
  if (!index) return;
  try {
    if (index->sharded) sharded_free(index->sharded);
    delete index->flat;
  } catch (...) {
  }
  delete index;
}

size_t VecSimIndex_EstimateElementSize(const VecSimParams *params) {
  if (!params || params->algo != VecSimAlgo_BF) return 0;
  const BFParams &bf = params->algoParams.bfParams;
  // one stored row + its label (host row_label entry and device label)
  return round_up(bf.dim * type_size(bf.type), 16) + 2 * sizeof(uint64_t);
}
size_t VecSimIndex_EstimateInitialSize(const VecSimParams *params) {
  if (!params || params->algo != VecSimAlgo_BF) return 0;
  return sizeof(FlatIndex) + sizeof(VecSimIndex);
}

// ---- writes --------------------------------------------------------------------------------------
int VecSimIndex_AddVector(VecSimIndex *index, const void *blob, size_t label) {
  if (!index || !blob) return 0;
  ABI_TRY
  if (index->sharded) return sharded_add(index->sharded, blob, label);
  return index->flat->add(blob, label);
  ABI_CATCH(log_ctx_of(index), "VecSimIndex_AddVector", 0)
}
int VecSimIndex_DeleteVector(VecSimIndex *index, size_t label) {
  if (!index) return 0;
  ABI_TRY
  if (index->sharded) return sharded_remove(index->sharded, label);
  return index->flat->remove(label);
  ABI_CATCH(log_ctx_of(index), "VecSimIndex_DeleteVector", 0)
}

// ---- info ----------------------------------------------------------------------------------------
size_t VecSimIndex_IndexSize(VecSimIndex *index) {
  if (!index) return 0;
  return index->sharded ? sharded_size(index->sharded) : index->flat->size();
}
VecSimIndexBasicInfo VecSimIndex_BasicInfo(VecSimIndex *index) {
  if (index) return (index->sharded ? sharded_first(index->sharded) : index->flat)->basic_info();
  VecSimIndexBasicInfo i;
  memset(&i, 0, sizeof i);
  return i;
}
VecSimIndexStatsInfo VecSimIndex_StatsInfo(VecSimIndex *index) {
  VecSimIndexStatsInfo s;
  memset(&s, 0, sizeof s);
  if (index) s.memory = index->sharded ? sharded_memory(index->sharded) : index->flat->memory();
  return s;
}

static const char *type_str(VecSimType t) {
  switch (t) {
    case VecSimType_FLOAT32: return "FLOAT32";
    case VecSimType_FLOAT64: return "FLOAT64";
    case VecSimType_BFLOAT16: return "BFLOAT16";
    case VecSimType_FLOAT16: return "FLOAT16";
    case VecSimType_INT8: return "INT8";
    case VecSimType_UINT8: return "UINT8";
    default: return "UNKNOWN";
  }
}
static const char *metric_str(VecSimMetric m) {
  return m == VecSimMetric_L2 ? "L2" : m == VecSimMetric_IP ? "IP" : "COSINE";
}
static const char *mode_str(int m) {
  switch (m) {
    case EMPTY_MODE: return "EMPTY_MODE";
    case STANDARD_KNN: return "STANDARD_KNN";
    case HYBRID_ADHOC_BF: return "HYBRID_ADHOC_BF";
    case HYBRID_BATCHES: return "HYBRID_BATCHES";
    case HYBRID_BATCHES_TO_ADHOC_BF: return "HYBRID_BATCHES_TO_ADHOC_BF";
    case RANGE_QUERY: return "RANGE_QUERY";
    default: return "UNKNOWN";
  }
}

// Field list of a FLAT index as FT.DEBUG VECSIM_INFO prints it (reference
// tests/pytests/test_vecsim.py:342, the FRONTEND_INDEX part).
VecSimDebugInfoIterator *VecSimIndex_DebugInfoIterator(VecSimIndex *index) {
  if (!index) return nullptr;
  RSGPU_ShardedIndex *sh = index->sharded;
  FlatIndex *f = sh ? sharded_first(sh) : index->flat;  // (a sharded handle: type / metric / dim from shard 0, sizes summed)
  auto *it = new VecSimDebugInfoIterator();
  auto add_s = [&](const char *n, const char *v) {
    VecSim_InfoField fld; fld.fieldName = n; fld.fieldType = INFOFIELD_STRING; fld.fieldValue.stringValue = v;
    it->fields.push_back(fld);
  };
  auto add_u = [&](const char *n, uint64_t v) {
    VecSim_InfoField fld; fld.fieldName = n; fld.fieldType = INFOFIELD_UINT64; fld.fieldValue.uintegerValue = v;
    it->fields.push_back(fld);
  };
  add_s("ALGORITHM", "FLAT");
  add_s("TYPE", type_str(f->type));
  add_u("DIMENSION", f->dim);
  add_s("METRIC", metric_str(f->metric));
  add_u("IS_MULTI_VALUE", f->multi);
  add_u("IS_DISK", 0);
  add_u("INDEX_SIZE", sh ? sharded_size(sh) : f->size());
  add_u("INDEX_LABEL_COUNT", sh ? sharded_label_count(sh) : f->label_count());
  add_u("MEMORY", sh ? sharded_memory(sh) : f->memory());
  add_s("LAST_SEARCH_MODE", mode_str(sh ? sharded_last_mode(sh) : f->last_mode.load()));
  add_u("BLOCK_SIZE", f->block_size);
  return it;
}
size_t VecSimDebugInfoIterator_NumberOfFields(VecSimDebugInfoIterator *it) { return it ? it->fields.size() : 0; }
bool VecSimDebugInfoIterator_HasNextField(VecSimDebugInfoIterator *it) { return it && it->pos < it->fields.size(); }
VecSim_InfoField *VecSimDebugInfoIterator_NextField(VecSimDebugInfoIterator *it) {
  return (it && it->pos < it->fields.size()) ? &it->fields[it->pos++] : nullptr;
}
void VecSimDebugInfoIterator_Free(VecSimDebugInfoIterator *it) { delete it; }

// ---- query-param resolution (host logic; pins: reference tests/pytests/test_vecsim.py:692-765) -------
static bool ieq(const VecSimRawParam &p, const char *name) {
  return p.nameLen == strlen(name) && strncasecmp(p.name, name, p.nameLen) == 0;
}
static bool parse_positive(const VecSimRawParam &p, size_t *out) {
  if (!p.valLen) return false;
  std::string s(p.value, p.valLen);
  char *end = nullptr;
  errno = 0;
  long long v = strtoll(s.c_str(), &end, 10);
  if (errno || end == s.c_str() || *end != '\0' || v <= 0) return false;
  *out = (size_t)v;
  return true;
}
VecSimResolveCode VecSimIndex_ResolveParams(VecSimIndex *index, VecSimRawParam *rparams, int paramNum,
                                            VecSimQueryParams *qparams, VecsimQueryType query_type) {
  (void)index;
  if (!qparams || (paramNum && !rparams)) return VecSimParamResolverErr_UnknownParam;
  void *tctx = qparams->timeoutCtx;
  memset(qparams, 0, sizeof *qparams);
  qparams->timeoutCtx = tctx;
  bool batch_set = false;
  for (int i = 0; i < paramNum; i++) {
    const VecSimRawParam &p = rparams[i];
    if (ieq(p, "EPSILON")) {
      // range-only, and only for approximate indexes: FLAT has no epsilon
      if (query_type != QUERY_TYPE_RANGE) return VecSimParamResolverErr_InvalidPolicy_NRange;
      return VecSimParamResolverErr_UnknownParam;
    } else if (ieq(p, "BATCH_SIZE")) {
      if (query_type != QUERY_TYPE_HYBRID) return VecSimParamResolverErr_InvalidPolicy_NHybrid;
      if (batch_set) return VecSimParamResolverErr_AlreadySet;
      size_t v;
      if (!parse_positive(p, &v)) return VecSimParamResolverErr_BadValue;
      qparams->batchSize = v;
      batch_set = true;
    } else if (ieq(p, "HYBRID_POLICY")) {
      if (query_type != QUERY_TYPE_HYBRID) return VecSimParamResolverErr_InvalidPolicy_NHybrid;
      if (qparams->searchMode != EMPTY_MODE) return VecSimParamResolverErr_AlreadySet;
      if (p.valLen == 7 && !strncasecmp(p.value, "batches", 7)) qparams->searchMode = HYBRID_BATCHES;
      else if (p.valLen == 8 && !strncasecmp(p.value, "adhoc_bf", 8)) qparams->searchMode = HYBRID_ADHOC_BF;
      else return VecSimParamResolverErr_InvalidPolicy_NExits;
    } else {
      // EF_RUNTIME, RERANK, SVS knobs ...: not options of a FLAT index
      return VecSimParamResolverErr_UnknownParam;
    }
  }
  if (qparams->searchMode == HYBRID_ADHOC_BF && qparams->batchSize > 0)
    return VecSimParamResolverErr_InvalidPolicy_AdHoc_With_BatchSize;
  return VecSim_OK;
}

// ---- queries -------------------------------------------------------------------------------------
VecSimQueryReply *VecSimIndex_TopKQuery(VecSimIndex *index, const void *queryBlob, size_t k,
                                        VecSimQueryParams *queryParams, VecSimQueryReply_Order order) {
  if (!index || !queryBlob) return nullptr;
  ABI_TRY
  if (index->sharded) return sharded_topk(index->sharded, queryBlob, k, queryParams, order);
  return index->flat->topk(queryBlob, k, queryParams, order);
  ABI_CATCH(log_ctx_of(index), "VecSimIndex_TopKQuery", nullptr)
}

VecSimQueryReply *VecSimIndex_RangeQuery(VecSimIndex *index, const void *queryBlob, double radius,
                                         VecSimQueryParams *queryParams, VecSimQueryReply_Order order) {
  if (!index || !queryBlob) return nullptr;
  ABI_TRY
  if (index->sharded) return sharded_range(index->sharded, queryBlob, radius, queryParams, order);
  return index->flat->range(queryBlob, radius, queryParams, order);
  ABI_CATCH(log_ctx_of(index), "VecSimIndex_RangeQuery", nullptr)
}

double VecSimIndex_GetDistanceFrom_Unsafe(VecSimIndex *index, size_t label, const void *blob) {
  if (!index || !blob) return NAN;
  ABI_TRY
  if (index->sharded) return sharded_distance_from(index->sharded, label, blob);
  return index->flat->distance_from(label, blob);
  ABI_CATCH(log_ctx_of(index), "VecSimIndex_GetDistanceFrom_Unsafe", NAN)
}

bool VecSimIndex_PreferAdHocSearch(VecSimIndex *index, size_t subsetSize, size_t k, bool initialCheck) {
  if (!index) return true;
  ABI_TRY
  if (index->sharded) return sharded_prefer_adhoc(index->sharded, subsetSize, k, initialCheck);
  return index->flat->prefer_adhoc(subsetSize, k, initialCheck);
  ABI_CATCH(log_ctx_of(index), "VecSimIndex_PreferAdHocSearch", true)
}

// ---- batch iterator ------------------------------------------------------------------------------
VecSimBatchIterator *VecSimBatchIterator_New(VecSimIndex *index, const void *queryBlob, VecSimQueryParams *queryParams) {
  if (!index || !queryBlob) return nullptr;
  if (index->sharded) {
    try {
      std::unique_ptr<VecSimBatchIterator> b(new VecSimBatchIterator());
      b->it.index = nullptr;
      b->it.ctx = nullptr;
      b->sh = sharded_batch_new(index->sharded, queryBlob, queryParams);
      return b.release();
    } catch (const std::exception &e) {
      set_error(log_ctx_of(index), "VecSimBatchIterator_New", e.what());
      return nullptr;
    } catch (...) {  // nothing may cross the C ABI
      set_error(log_ctx_of(index), "VecSimBatchIterator_New", "unknown error");
      return nullptr;
    }
  }
  FlatIndex *f = index->flat;
  ABI_TRY
  f->flush_if_needed();
  HIP_CHECK(hipSetDevice(f->device));
  std::unique_ptr<VecSimBatchIterator> b(new VecSimBatchIterator());
  b->it.index = f;
  size_t bytes = f->dim * type_size(f->type);
  b->it.query.assign((const uint8_t *)queryBlob, (const uint8_t *)queryBlob + bytes);
  b->it.timeout_ctx = queryParams ? queryParams->timeoutCtx : nullptr;
  b->it.n = f->committed_rows();
  b->it.ctx = CtxPool::get().acquire(f->device);  // (last: nothing after it can throw and strand the leased context)
  return b.release();
  ABI_CATCH(f->log_ctx, "VecSimBatchIterator_New", nullptr)
}

bool VecSimBatchIterator_HasNext(VecSimBatchIterator *iterator) {
  if (!iterator) return false;
  if (iterator->sh) return sharded_batch_has_next(iterator->sh);
  return iterator->it.returned < iterator->it.n;
}

VecSimQueryReply *VecSimBatchIterator_Next(VecSimBatchIterator *iterator, size_t n_results, VecSimQueryReply_Order order) {
  if (!iterator) return nullptr;
  if (iterator->sh) {
    try {
      return sharded_batch_next(iterator->sh, n_results, order);
    } catch (const std::exception &e) {
      set_error(nullptr, "VecSimBatchIterator_Next", e.what());
      return nullptr;
    } catch (...) {
      set_error(nullptr, "VecSimBatchIterator_Next", "unknown error");
      return nullptr;
    }
  }
  BatchIterator &b = iterator->it;
  FlatIndex *f = b.index;
  ABI_TRY
  f->last_mode = HYBRID_BATCHES;
  if (timed_out(b.timeout_ctx)) return new_reply(0, VecSim_QueryReply_TimedOut);
  std::shared_lock<std::shared_mutex> g(f->mu);
  HIP_CHECK(hipSetDevice(f->device));
  uint32_t n = std::min<uint32_t>(b.n, f->committed_rows());
  // A DeleteVector between two Next() calls moves rows (swap-delete): the keys of the first scan would pair a moved
  // row's old score with another document's label.  The caller's lock normally excludes that (SURVEY.md 8b
  // threading contract); if it happened anyway the keys are recomputed for the current layout and the iterator
  // continues above the same (key,row) bound.
  if (!b.scanned || b.epoch != f->layout_epoch.load()) {
    if (b.scanned) {  // some of the rows already yielded may be gone: count down by short selections from now on
      b.recount = true;
      if (n && b.returned >= n) b.returned = n - 1;
    }
    b.n = n;
    f->upload_query(b.ctx, b.query.data(), true);
    f->scan_all(b.ctx, n);
    b.scanned = true;
    b.epoch = f->layout_epoch.load();
  }
  std::vector<VecSimQueryResult> res;
  std::vector<Hit> hits;
  size_t want = n_results;
  while (res.size() < want && b.returned < n) {
    uint32_t ask = (uint32_t)std::min<size_t>(b.recount ? n : n - b.returned, want - res.size());
    Bound bound;
    f->select(b.ctx, n, ask, b.lower, hits, &bound);
    if (hits.empty()) { b.returned = n; break; }
    if (b.recount) b.returned = hits.size() < ask ? n : std::min<uint32_t>(b.returned + (uint32_t)hits.size(), n - 1);
    else b.returned += (uint32_t)hits.size();
    b.lower = bound;
    for (const Hit &h : hits) {
      uint64_t lab = f->label_of_row(h.row);
      if (f->multi) {  // a label is yielded once, at its best vector
        if (!b.seen_labels.insert(lab).second) continue;
      }
      res.push_back(VecSimQueryResult{(size_t)lab, f->score_of(h.key)});
    }
    if (!f->multi) break;
    if (timed_out(b.timeout_ctx)) return new_reply(0, VecSim_QueryReply_TimedOut);
  }
  VecSimQueryReply *r = new_reply(res.size(), VecSim_QueryReply_OK);
  if (!res.empty()) memcpy(r->results, res.data(), res.size() * sizeof(VecSimQueryResult));
  if (order == BY_ID)
    std::sort(r->results, r->results + r->len, [](const VecSimQueryResult &a, const VecSimQueryResult &c) { return a.id < c.id; });
  else
    std::sort(r->results, r->results + r->len, [](const VecSimQueryResult &a, const VecSimQueryResult &c) {
      return score_id_before(a.score, a.id, c.score, c.id);
    });
  return r;
  ABI_CATCH(f->log_ctx, "VecSimBatchIterator_Next", nullptr)
}

void VecSimBatchIterator_Reset(VecSimBatchIterator *iterator) {
  if (!iterator) return;
  if (iterator->sh) {
    sharded_batch_reset(iterator->sh);
    return;
  }
  iterator->it.returned = 0;
  iterator->it.recount = false;
  iterator->it.lower = Bound();
  iterator->it.seen_labels.clear();
}

void VecSimBatchIterator_Free(VecSimBatchIterator *iterator) {
  if (!iterator) return;
  if (iterator->sh) sharded_batch_free(iterator->sh);
  if (iterator->it.ctx) CtxPool::get().release(iterator->it.ctx);
  delete iterator;
}

// ---- ad-hoc brute-force context (batched GPU gather) -----------------------------------------------
VecSimAdhocBfCtx *VecSimIndex_AdhocBfCtx_New(VecSimIndex *index, const void *queryBlob) {
  if (!index || !queryBlob) return nullptr;
  if (index->sharded) {
    try {
      auto *a = new VecSimAdhocBfCtx();
      a->a.index = nullptr;
      a->a.ctx = nullptr;
      a->sh = sharded_adhoc_new(index->sharded, queryBlob);
      return a;
    } catch (const std::exception &e) {
      set_error(log_ctx_of(index), "VecSimIndex_AdhocBfCtx_New", e.what());
      return nullptr;
    } catch (...) {
      set_error(log_ctx_of(index), "VecSimIndex_AdhocBfCtx_New", "unknown error");
      return nullptr;
    }
  }
  FlatIndex *f = index->flat;
  ABI_TRY
  f->flush_if_needed();
  HIP_CHECK(hipSetDevice(f->device));
  auto *a = new VecSimAdhocBfCtx();
  a->a.index = f;
  a->a.ctx = CtxPool::get().acquire(f->device);
  f->upload_query(a->a.ctx, queryBlob, true);  // normalises a copy for cosine (hybrid_reader.c:212-214)
  return a;
  ABI_CATCH(f->log_ctx, "VecSimIndex_AdhocBfCtx_New", nullptr)
}
void VecSimIndex_AdhocBfCtx_GetExactDistances(VecSimAdhocBfCtx *ctx, const size_t *labels, double *out, size_t count) {
  if (!ctx || !count) return;
  if (ctx->sh) {
    try {
      sharded_adhoc_distances(ctx->sh, labels, out, count);
    } catch (const std::exception &e) {
      set_error(nullptr, "VecSimIndex_AdhocBfCtx_GetExactDistances", e.what());
      for (size_t i = 0; i < count; i++) out[i] = NAN;
    } catch (...) {
      set_error(nullptr, "VecSimIndex_AdhocBfCtx_GetExactDistances", "unknown error");
      for (size_t i = 0; i < count; i++) out[i] = NAN;
    }
    return;
  }
  FlatIndex *f = ctx->a.index;
  try {
    std::shared_lock<std::shared_mutex> g(f->mu);
    HIP_CHECK(hipSetDevice(f->device));
    f->gather(ctx->a.ctx, labels, count, out);
  } catch (const std::exception &e) {
    set_error(f->log_ctx, "VecSimIndex_AdhocBfCtx_GetExactDistances", e.what());
    for (size_t i = 0; i < count; i++) out[i] = NAN;
  } catch (...) {
    set_error(f->log_ctx, "VecSimIndex_AdhocBfCtx_GetExactDistances", "unknown error");
    for (size_t i = 0; i < count; i++) out[i] = NAN;
  }
}
double VecSimIndex_AdhocBfCtx_GetDistanceFrom(VecSimAdhocBfCtx *ctx, size_t label) {
  double d = NAN;
  VecSimIndex_AdhocBfCtx_GetExactDistances(ctx, &label, &d, 1);
  return d;
}
void VecSimIndex_AdhocBfCtx_Free(VecSimAdhocBfCtx *ctx) {
  if (!ctx) return;
  if (ctx->sh) sharded_adhoc_free(ctx->sh);
  if (ctx->a.ctx) CtxPool::get().release(ctx->a.ctx);
  delete ctx;
}

void VecSimTieredIndex_AcquireSharedLocks(VecSimIndex *) {}
void VecSimTieredIndex_ReleaseSharedLocks(VecSimIndex *) {}
void VecSimTieredIndex_GC(VecSimIndex *) {}
int VecSimDebug_GetElementNeighborsInHNSWGraph(VecSimIndex *, size_t, int ***neighborsData) {
  if (neighborsData) *neighborsData = nullptr;
  return VecSimDebugCommandCode_BadIndex;
}
void VecSimDebug_ReleaseElementNeighborsInHNSWGraph(int **) {}

// ---- blob helpers ----------------------------------------------------------------------------------
void VecSim_Normalize(void *blob, size_t dim, VecSimType type) {
  if (blob) normalize_blob(blob, dim, type);
}
size_t VecSimParams_GetQueryBlobSize(VecSimType type, size_t dim, VecSimMetric metric) {
  size_t s = dim * type_size(type);
  if (metric == VecSimMetric_Cosine && (type == VecSimType_INT8 || type == VecSimType_UINT8)) s += sizeof(float);
  return s;
}

// ---- process-wide hooks ----------------------------------------------------------------------------
void VecSim_SetMemoryFunctions(VecSimMemoryFunctions f) {
  if (f.allocFunction && f.callocFunction && f.reallocFunction && f.freeFunction) hooks().mem = f;
}
void VecSim_SetTimeoutCallbackFunction(timeoutCallbackFunction cb) { hooks().timeout = cb; }
void VecSim_SetLogCallbackFunction(logCallbackFunction cb) { hooks().log = cb; }
void VecSim_SetWriteMode(VecSimWriteMode) {}
void VecSim_UpdateThreadPoolSize(size_t n) { hooks().thread_pool_size = n; }
size_t VecSim_GetSharedMemory(void) { return CtxPool::get().bytes(); }

// ---- reply accessors ---------------------------------------------------------------------------------
size_t VecSimQueryResult_GetId(const VecSimQueryResult *item) { return item ? item->id : (size_t)-1; }
double VecSimQueryResult_GetScore(const VecSimQueryResult *item) { return item ? item->score : NAN; }
size_t VecSimQueryReply_Len(VecSimQueryReply *reply) { return reply ? reply->len : 0; }
VecSimQueryReply_Code VecSimQueryReply_GetCode(VecSimQueryReply *reply) {
  return reply ? reply->code : VecSim_QueryReply_OK;
}
void VecSimQueryReply_Free(VecSimQueryReply *reply) {
  if (!reply) return;
  host_free(reply->results);
  host_free(reply);
}
VecSimQueryReply_Iterator *VecSimQueryReply_GetIterator(VecSimQueryReply *reply) {
  if (!reply) return nullptr;
  VecSimQueryReply_Iterator *it = host_alloc<VecSimQueryReply_Iterator>(1);
  it->reply = reply;
  it->pos = 0;
  return it;
}
VecSimQueryResult *VecSimQueryReply_IteratorNext(VecSimQueryReply_Iterator *it) {
  if (!it || it->pos >= it->reply->len) return nullptr;
  return &it->reply->results[it->pos++];
}
bool VecSimQueryReply_IteratorHasNext(VecSimQueryReply_Iterator *it) { return it && it->pos < it->reply->len; }
void VecSimQueryReply_IteratorReset(VecSimQueryReply_Iterator *it) {
  if (it) it->pos = 0;
}
void VecSimQueryReply_IteratorFree(VecSimQueryReply_Iterator *it) { host_free(it); }

// ---- extensions (include/rsgpu_ext.h) ------------------------------------------------------------------
const char *RSGPU_LastError(void) { return last_error().c_str(); }
int RSGPU_DeviceCount(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
int RSGPU_FlatIndex_Reserve(VecSimIndex *index, size_t rows) {
  if (!index) return -1;
  ABI_TRY
  if (index->sharded) sharded_reserve(index->sharded, rows);
  else index->flat->reserve(rows);
  return 0;
  ABI_CATCH(log_ctx_of(index), "RSGPU_FlatIndex_Reserve", -1)
}
int RSGPU_FlatIndex_AddDeviceRows(VecSimIndex *index, const void *dev_rows, size_t n, size_t first_label) {
  if (!index) return -1;
  ABI_TRY
  if (index->sharded) throw std::runtime_error("device rows belong to one device: load the shards one by one (RSGPU_ShardedIndex_Shard)");
  return index->flat->add_device_rows(dev_rows, n, first_label);
  ABI_CATCH(log_ctx_of(index), "RSGPU_FlatIndex_AddDeviceRows", -1)
}
int RSGPU_FlatIndex_ReadRows(VecSimIndex *index, size_t row_begin, size_t n, void *host_out) {
  if (!index || (n && !host_out)) return -1;
  ABI_TRY
  if (index->sharded) throw std::runtime_error("storage rows are per shard (RSGPU_ShardedIndex_Shard)");
  index->flat->read_rows((uint32_t)row_begin, n, host_out);
  return 0;
  ABI_CATCH(log_ctx_of(index), "RSGPU_FlatIndex_ReadRows", -1)
}
long RSGPU_FlatIndex_AddPhiloxRows(VecSimIndex *index, uint64_t seed, uint64_t first_index, size_t n, size_t first_label) {
  if (!index) return -1;
  ABI_TRY
  if (index->sharded) return sharded_add_philox_rows(index->sharded, seed, first_index, n, first_label);
  return index->flat->add_philox_rows(seed, first_index, n, first_label);
  ABI_CATCH(log_ctx_of(index), "RSGPU_FlatIndex_AddPhiloxRows", -1)
}
int RSGPU_FlatIndex_LabelTable(VecSimIndex *index) {
  if (!index || !index->flat) return -1;
  std::shared_lock<std::shared_mutex> g(index->flat->mu);
  return index->flat->label_mode();
}
int RSGPU_FlatIndex_TopKDevice(VecSimIndex *index, const void *query, size_t k, float *dev_scores, uint64_t *dev_labels) {
  if (!index || !query || !k) return -1;
  if (index->sharded) {
    set_error(log_ctx_of(index), "RSGPU_FlatIndex_TopKDevice", "a sharded handle has no single device to write to");
    return -1;
  }
  FlatIndex *f = index->flat;
  ABI_TRY
  VecSimQueryReply *r = f->topk(query, k, nullptr, BY_SCORE);
  std::vector<float> sc(k, INFINITY);
  std::vector<uint64_t> lb(k, UINT64_MAX);
  int got = (int)r->len;
  for (size_t i = 0; i < r->len; i++) {
    sc[i] = (float)r->results[i].score;
    lb[i] = r->results[i].id;
  }
  VecSimQueryReply_Free(r);
  HIP_CHECK(hipSetDevice(f->device));
  HIP_CHECK(hipMemcpy(dev_scores, sc.data(), k * sizeof(float), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dev_labels, lb.data(), k * sizeof(uint64_t), hipMemcpyHostToDevice));
  return got;
  ABI_CATCH(f->log_ctx, "RSGPU_FlatIndex_TopKDevice", -1)
}
int RSGPU_FlatIndex_TopKBatch(VecSimIndex *index, const void *queries, size_t n_queries, size_t k, size_t *ids_out,
                              double *scores_out, size_t *counts_out) {
  if (!index || !queries || !ids_out || !scores_out || !counts_out) return -1;
  ABI_TRY
  if (index->sharded) throw std::runtime_error("batched queries run per shard (RSGPU_ShardedIndex_Shard) and merge on the caller's side");
  index->flat->topk_batch(queries, n_queries, k, ids_out, scores_out, counts_out);
  return 0;
  ABI_CATCH(log_ctx_of(index), "RSGPU_FlatIndex_TopKBatch", -1)
}
// the coordinator-style K-way merge of per-shard top-k lists (reference src/module.c:3541-3547): k best of m
// (score,label) candidates by (score, label) ascending; padding slots carry label == UINT64_MAX.  Pure host code.
int RSGPU_MergeTopKHost(const float *scores, const uint64_t *labels, size_t m, size_t k, double *scores_out,
                        uint64_t *labels_out) {
  if ((m && (!scores || !labels)) || (k && (!scores_out || !labels_out))) return -1;
  ABI_TRY
  std::vector<size_t> ord;
  ord.reserve(m);
  for (size_t i = 0; i < m; i++)
    if (labels[i] != UINT64_MAX) ord.push_back(i);
  size_t kk = std::min(k, ord.size());
  std::partial_sort(ord.begin(), ord.begin() + (long)kk, ord.end(), [&](size_t a, size_t b) {
    return score_id_before(scores[a], labels[a], scores[b], labels[b]);
  });
  for (size_t i = 0; i < kk; i++) {
    scores_out[i] = (double)scores[ord[i]];
    labels_out[i] = labels[ord[i]];
  }
  return (int)kk;
  ABI_CATCH(nullptr, "RSGPU_MergeTopKHost", -1)
}
int RSGPU_MergeTopK(int device, const float *dev_scores, const uint64_t *dev_labels, size_t m, size_t k,
                    double *scores_out, uint64_t *labels_out, void *wait_stream) {
  ABI_TRY
  HIP_CHECK(hipSetDevice(device));
  if (wait_stream) HIP_CHECK(hipStreamSynchronize((hipStream_t)wait_stream));
  std::vector<float> sc(m);
  std::vector<uint64_t> lb(m);
  HIP_CHECK(hipMemcpy(sc.data(), dev_scores, m * sizeof(float), hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(lb.data(), dev_labels, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return RSGPU_MergeTopKHost(sc.data(), lb.data(), m, k, scores_out, labels_out);
  ABI_CATCH(nullptr, "RSGPU_MergeTopK", -1)
}

void RSGPU_SetProfiling(int on) { scan_profile().enabled = on; }
void RSGPU_ResetProfile(void) {
  scan_profile().launches = 0;
  scan_profile().bytes = 0;
  scan_profile().nanos = 0;
}
void RSGPU_GetScanProfile(uint64_t *launches, double *total_ms, uint64_t *bytes) {
  if (launches) *launches = scan_profile().launches.load();
  if (total_ms) *total_ms = (double)scan_profile().nanos.load() / 1e6;
  if (bytes) *bytes = scan_profile().bytes.load();
}
int RSGPU_LastBatchRoute(void) { return last_batch_route(); }
void RSGPU_GetCoalesceStats(uint64_t out[8]) {
  if (!out) return;
  CoalesceStats &c = coalesce_stats();
  const uint64_t v[8] = {c.passes.load(), c.queries.load(), c.mq_passes.load(), c.mq_queries.load(),
                         c.lingers.load(), c.linger_ns.load(), c.mq_device_ns.load(), c.mq_redo.load()};
  memcpy(out, v, sizeof v);
}
void RSGPU_ResetCoalesceStats(void) {
  CoalesceStats &c = coalesce_stats();
  c.passes = c.queries = c.mq_passes = c.mq_queries = c.lingers = c.linger_ns = c.mq_device_ns = c.mq_redo = 0;
  c.wide_passes = c.wide_queries = c.left_queue = 0;
}
void RSGPU_GetWidePassStats(uint64_t out[2]) {
  if (!out) return;
  out[0] = coalesce_stats().wide_passes.load();
  out[1] = coalesce_stats().wide_queries.load();
}
uint64_t RSGPU_GetCoalesceTimeouts(void) { return coalesce_stats().left_queue.load(); }
const char *RSGPU_GetLastMqScanKernel(char *buf, size_t cap) {
  if (!buf || !cap) return "";
  return last_scan_mq_kernel_name(buf, cap);
}
void RSGPU_GetTwoStageStats(uint64_t out[8]) {
  if (!out) return;
  for (int i = 0; i < 8; i++) out[i] = i < TwoStageStats::N ? two_stage_stats().v[i].load() : 0;
}
void RSGPU_ResetTwoStageStats(void) {
  for (auto &x : two_stage_stats().v) x = 0;
}
const char *RSGPU_GetLastScanKernel(char *buf, size_t cap) {
  if (!buf || !cap) return "";
  return last_scan_kernel_name(buf, cap);
}
int RSGPU_SetTuning(const char *key, int value) {
  if (!key) return -1;
  if (!strcmp(key, "blocks_per_cu")) scan_tuning().blocks_per_cu = value;
  else if (!strcmp(key, "rows_per_group")) scan_tuning().rows_per_group = value;
  else if (!strcmp(key, "nontemporal")) scan_tuning().nontemporal = value;
  else if (!strcmp(key, "gemm_dma")) scan_tuning().gemm_dma = value;
  else if (!strcmp(key, "filter_select")) scan_tuning().filter_select = value;
  else if (!strcmp(key, "gemm_qs")) scan_tuning().gemm_qs = value;
  else if (!strcmp(key, "gemm_qs_f32")) scan_tuning().gemm_qs_f32 = value;
  else if (!strcmp(key, "gemm_qs_h8")) scan_tuning().gemm_qs_h8 = value;
  else if (!strcmp(key, "gemm_qs_f8")) scan_tuning().gemm_qs_f8 = value;
  else if (!strcmp(key, "batch_prune")) scan_tuning().batch_prune = value;
  else if (!strcmp(key, "batch_select_regs")) scan_tuning().batch_select_regs = value;
  else if (!strcmp(key, "hybrid_coalesce")) scan_tuning().hybrid_coalesce = value;
  else if (!strcmp(key, "hybrid_coalesce_depth")) scan_tuning().hybrid_coalesce_depth = value;
  else if (!strcmp(key, "hybrid_coalesce_interleave")) scan_tuning().hybrid_coalesce_interleave = value;
  else if (!strcmp(key, "shard_exchange")) scan_tuning().shard_exchange = value;
  else if (!strcmp(key, "hybrid_dir")) scan_tuning().hybrid_dir = value;
  else if (!strcmp(key, "hybrid_poll")) scan_tuning().hybrid_poll = value;
  else if (!strcmp(key, "hybrid_select_split")) scan_tuning().hybrid_select_split = value;
  else if (!strcmp(key, "hybrid_knn_pipeline")) scan_tuning().hybrid_knn_pipeline = value;
  else if (!strcmp(key, "hybrid_packed_docs")) scan_tuning().hybrid_packed_docs = value;
  else if (!strcmp(key, "qs_phases")) scan_tuning().qs_phases = value;
  else if (!strcmp(key, "qs_force_i8")) scan_tuning().qs_force_i8 = value;
  else if (!strcmp(key, "shards")) scan_tuning().shards = value;
  else if (!strcmp(key, "shard_replicas")) scan_tuning().shard_replicas = value;
  else if (!strcmp(key, "cache_decoded")) scan_tuning().cache_decoded = value;
  else if (!strcmp(key, "decode_dense")) scan_tuning().decode_dense = value;
  else if (!strcmp(key, "decode_sync")) scan_tuning().decode_sync = value;
  else if (!strcmp(key, "probe_dpt")) scan_tuning().probe_dpt = value;
  else if (!strcmp(key, "decode_pair")) scan_tuning().decode_pair = value;
  else if (!strcmp(key, "hybrid_tiles")) scan_tuning().hybrid_tiles = value;
  else if (!strcmp(key, "hybrid_tree_tiles")) scan_tuning().hybrid_tree_tiles = value;
  else if (!strcmp(key, "hybrid_force_general")) scan_tuning().hybrid_force_general = value;
  else if (!strcmp(key, "prioritize_union_children")) scan_tuning().prioritize_union_children = value;
  else if (!strcmp(key, "hybrid_surv_cap")) scan_tuning().hybrid_surv_cap = value;
  else if (!strcmp(key, "hybrid_trace")) scan_tuning().hybrid_trace = value;
  else if (!strcmp(key, "decode_lean")) scan_tuning().decode_lean = value;
  else if (!strcmp(key, "mq16")) scan_tuning().mq16 = value;
  else if (!strcmp(key, "coalesce_shadow8")) scan_tuning().coalesce_shadow8 = value;
  else if (!strcmp(key, "shadow16")) scan_tuning().shadow16 = value;
  else if (!strcmp(key, "two_stage")) scan_tuning().two_stage = value;
  else if (!strcmp(key, "shadow8")) scan_tuning().shadow8 = value;
  else if (!strcmp(key, "coalesce")) scan_tuning().coalesce = value;
  else if (!strcmp(key, "coalesce_linger_us")) scan_tuning().coalesce_linger_us = value;
  else if (!strcmp(key, "coalesce_wide")) scan_tuning().coalesce_wide = value;
  else if (!strcmp(key, "coalesce_wide_min")) scan_tuning().coalesce_wide_min = value;
  else if (!strcmp(key, "coalesce_min_mib")) scan_tuning().coalesce_min_mib = value;
  else if (!strcmp(key, "mq_blocks_per_cu")) scan_tuning().mq_blocks_per_cu = value;
  else if (!strcmp(key, "batch_mfma")) scan_tuning().batch_mfma = value;
  else if (!strcmp(key, "vmm")) scan_tuning().vmm = value;
  else if (!strcmp(key, "vmm_chunk_mib")) scan_tuning().vmm_chunk_mib = value;
  else if (!strcmp(key, "vmm_reserve_factor")) scan_tuning().vmm_reserve_factor = value > 0 ? value : 64;
  else return -1;
  return 0;
}
void RSGPU_ReleaseWorkspaces(void) {
  CtxPool::get().drain();
  rsgpu::release_search_pool();  // parked hit-list buffers of the search seam
  rsgpu::release_batch_pool();   // the batched path's scratch
}

}  // extern "C"
