/*
 * VecSim/vec_sim_common.h -- types of the VecSim C ABI, reconstructed for the MI355X FLAT engine.
 *
 * This header REPLACES deps/VectorSimilarity/src/VecSim/vec_sim_common.h (an un-vendored,
 * empty submodule in the reference tree, /root/reference/.gitmodules:10-12).  It declares every
 * type, enumerator and struct field that RediSearch's C code touches at the VecSim seam, so the
 * reference's callers compile against it unchanged.  Each declaration cites the reference call
 * site that pins it; what the reference does not pin is marked [upstream-memory] (SURVEY.md
 * Appendix A / D).
 *
 * Numeric enum values are an on-disk contract: VecSim_RdbSave writes algo/type/metric as
 * unsigned ints (reference src/vector_index.c:486-495).
 */
#ifndef VECSIM_VEC_SIM_COMMON_H
#define VECSIM_VEC_SIM_COMMON_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
#include <limits.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- defaults (reference src/config.h:386 VECSIM_DEFAULT_BLOCK_SIZE) ---------------------- */
#define DEFAULT_BLOCK_SIZE 1024
#define HNSW_DEFAULT_M 16
#define HNSW_DEFAULT_EF_C 200
#define HNSW_DEFAULT_EF_RT 10
#define HNSW_DEFAULT_EPSILON 0.01
#define VECSIM_SVS_DEFAULT_EPSILON 0.01
#define HNSW_INVALID_LEVEL SIZE_MAX
#define INVALID_JOB_ID UINT_MAX
#define INVALID_INFO UINT_MAX

/* SVS-Vamana creation defaults.  Not served by this engine; declared because the FT.CREATE parser fills
 * them in before it knows which algorithm it will get (reference src/spec.c:1227-1237).  Values pinned by
 * the FT.DEBUG VECSIM_INFO expectation at reference tests/pytests/test_vecsim.py:357-360 (graph degree 32,
 * construction window 200, leanvec 0, alpha 1.2 / 0.95); the training threshold is 10 blocks
 * [upstream-memory]. */
#define SVS_VAMANA_DEFAULT_ALPHA_L2 1.2f
#define SVS_VAMANA_DEFAULT_ALPHA_IP 0.95f
#define SVS_VAMANA_DEFAULT_GRAPH_MAX_DEGREE 32
#define SVS_VAMANA_DEFAULT_CONSTRUCTION_WINDOW_SIZE 200
#define SVS_VAMANA_DEFAULT_USE_SEARCH_HISTORY true
#define SVS_VAMANA_DEFAULT_NUM_THREADS 1
#define SVS_VAMANA_DEFAULT_TRAINING_THRESHOLD (10 * DEFAULT_BLOCK_SIZE)
#define SVS_VAMANA_DEFAULT_UPDATE_THRESHOLD (1 * DEFAULT_BLOCK_SIZE)
#define SVS_VAMANA_DEFAULT_SEARCH_WINDOW_SIZE 10
#define SVS_VAMANA_DEFAULT_LEANVEC_DIM 0
#define SVS_VAMANA_DEFAULT_EPSILON 0.01f

/* HYBRID_POLICY values as VecSimIndex_ResolveParams accepts them (case-insensitive); the FT.HYBRID FILTER
 * parser maps its "ADHOC" onto the first (reference src/hybrid/parse/hybrid_callbacks.c:441-443). */
#define VECSIM_POLICY_ADHOC_BF "adhoc_bf"
#define VECSIM_POLICY_BATCHES "batches"

/* Element type of stored vectors and query blobs.
 * Names: reference src/vector_index.c:386-393; order [upstream-memory], RDB-persisted (:491). */
typedef enum {
  VecSimType_FLOAT32,
  VecSimType_FLOAT64,
  VecSimType_BFLOAT16,
  VecSimType_FLOAT16,
  VecSimType_INT8,
  VecSimType_UINT8,
  VecSimType_INT32,
  VecSimType_INT64
} VecSimType;

/* Index algorithm. Names: reference src/vector_index.c:423-426; persisted :487.
 * Only VecSimAlgo_BF (FLAT) is implemented by this library; the rest make VecSimIndex_New
 * return NULL. */
typedef enum { VecSimAlgo_BF, VecSimAlgo_HNSWLIB, VecSimAlgo_TIERED, VecSimAlgo_SVS } VecSimAlgo;

/* Distance metric. Names: reference src/vector_index.c:414-416; persisted :493. */
typedef enum { VecSimMetric_L2, VecSimMetric_IP, VecSimMetric_Cosine } VecSimMetric;

typedef enum { VecSimOption_AUTO = 0, VecSimOption_ENABLE = 1, VecSimOption_DISABLE = 2 } VecSimOptionMode;

/* Tri-state used by the disk runtime params (reference src/vector_index.c:268-270). */
typedef enum { VecSimBool_TRUE = 1, VecSimBool_FALSE = 0, VecSimBool_UNSET = -1 } VecSimBool;

typedef size_t labelType; /* == t_docId; reference src/document.c:721 passes the doc id */
typedef unsigned int idType;

/* Search mode recorded per query; mirrored 1:1 by reference src/vector_index.h:130-143. */
typedef enum {
  EMPTY_MODE,
  STANDARD_KNN,
  HYBRID_ADHOC_BF,
  HYBRID_BATCHES,
  HYBRID_BATCHES_TO_ADHOC_BF,
  RANGE_QUERY
} VecSearchMode;

/* What kind of query a param set is resolved for (reference src/vector_index.c:260,308). */
typedef enum { QUERY_TYPE_NONE, QUERY_TYPE_KNN, QUERY_TYPE_HYBRID, QUERY_TYPE_RANGE } VecsimQueryType;

/* Param-resolution result; names reference src/vector_index.c:751-781. */
typedef enum {
  VecSim_OK = 0,
  VecSimParamResolverErr_AlreadySet,
  VecSimParamResolverErr_UnknownParam,
  VecSimParamResolverErr_BadValue,
  VecSimParamResolverErr_InvalidPolicy_NExits,
  VecSimParamResolverErr_InvalidPolicy_NHybrid,
  VecSimParamResolverErr_InvalidPolicy_NRange,
  VecSimParamResolverErr_InvalidPolicy_AdHoc_With_BatchSize,
  VecSimParamResolverErr_InvalidPolicy_AdHoc_With_EfRuntime
} VecSimResolveCode;

/* Reply status (reference src/iterators/hybrid_reader.c:209,221,376). */
typedef enum {
  VecSim_QueryReply_OK = VecSim_OK,
  VecSim_QueryReply_TimedOut
} VecSimQueryReply_Code;

typedef enum { VecSim_WriteAsync, VecSim_WriteInPlace } VecSimWriteMode;

typedef enum {
  VecSimDebugCommandCode_OK = 0,
  VecSimDebugCommandCode_BadIndex,
  VecSimDebugCommandCode_LabelNotExists,
  VecSimDebugCommandCode_MultiNotSupported
} VecSimDebugCommandCode;

/* SVS quantisation flavours -- declared only so spec.c compiles (reference src/spec.c:563-581). */
typedef enum {
  VecSimSvsQuant_NONE = 0,
  VecSimSvsQuant_Scalar = 1,
  VecSimSvsQuant_4 = 4,
  VecSimSvsQuant_8 = 8,
  VecSimSvsQuant_4x4 = 4 | (4 << 8),
  VecSimSvsQuant_4x8 = 4 | (8 << 8),
  VecSimSvsQuant_4x8_LeanVec = 4 | (8 << 8) | (1 << 16),
  VecSimSvsQuant_8x8_LeanVec = 8 | (8 << 8) | (1 << 16)
} VecSimSvsQuantBits;

/* ---- creation params ------------------------------------------------------------------------ */

/* FLAT params; field order pinned by reference
 * src/redisearch_rs/query_eval/tests/integration/vector.rs:181-190. */
typedef struct {
  VecSimType type;
  size_t dim;
  VecSimMetric metric;
  bool multi;             /* several vectors per label (JSON multi-value, document.c:719-723) */
  size_t initialCapacity; /* deprecated upstream; used here as a reserve hint */
  size_t blockSize;       /* rows per block; reference src/spec.c:611-630 forces min(1024, limit) */
} BFParams;

/* HNSW params; field order pinned by reference
 * src/redisearch_rs/vector_score_source/src/test_utils.rs:80-93. Not implemented here. */
typedef struct {
  VecSimType type;
  size_t dim;
  VecSimMetric metric;
  bool multi;
  size_t initialCapacity;
  size_t blockSize;
  size_t M;
  size_t efConstruction;
  size_t efRuntime;
  double epsilon;
} HNSWParams;

/* SVS params; fields used by reference src/spec.c:915-1021. Not implemented here. */
typedef struct {
  VecSimType type;
  size_t dim;
  VecSimMetric metric;
  bool multi;
  size_t blockSize;
  VecSimSvsQuantBits quantBits;
  float alpha;
  size_t graph_max_degree;
  size_t construction_window_size;
  size_t max_candidate_pool_size;
  size_t prune_to;
  VecSimOptionMode use_search_history;
  size_t num_threads;
  size_t search_window_size;
  size_t search_buffer_capacity;
  size_t leanvec_dim;
  double epsilon;
} SVSParams;

typedef struct AsyncJob AsyncJob;
typedef int (*SubmitCB)(void *job_queue, void *index_ctx, AsyncJob **jobs, void **cbs, size_t jobs_len);
typedef int (*ThrottleCB)(void);

typedef struct {
  size_t swapJobThreshold;
} TieredHNSWParams;

typedef struct {
  size_t trainingTriggerThreshold; /* reference src/spec.c (TRAINING_THRESHOLD) */
  size_t updateTriggerThreshold;
  size_t updateJobWaitTime;
} TieredSVSParams;

typedef struct VecSimParams VecSimParams;

/* Tiered wrapper; fields used by reference src/vector_index.c VecSim_TieredParams_Init. */
typedef struct {
  void *jobQueue;
  void *jobQueueCtx;
  SubmitCB submitCb;
  size_t flatBufferLimit;
  VecSimParams *primaryIndexParams;
  union {
    TieredHNSWParams tieredHnswParams;
    TieredSVSParams tieredSVSParams;
  } specificParams;
} TieredIndexParams;

typedef union {
  HNSWParams hnswParams;
  BFParams bfParams;
  TieredIndexParams tieredParams;
  SVSParams svsParams;
} AlgoParams;

/* reference src/redisearch_rs/query_eval/tests/integration/vector.rs:179-192 */
struct VecSimParams {
  VecSimAlgo algo;
  AlgoParams algoParams;
  void *logCtx; /* passed back verbatim to the log callback (reference src/vector_index.c:803) */
};

/* Disk-backed index creation (Redis Flex). Declared for source compatibility only. */
typedef struct {
  const char *indexName;
  size_t indexNameLen;
  void *storage;
  void *userData; /* the owning field index; key of the disk layer (reference src/spec.c:1203,2920) */
  bool rerank;
} VecSimDiskContext;

typedef struct {
  VecSimParams *indexParams;
  VecSimDiskContext *diskContext;
} VecSimParamsDisk;

/* ---- query-time params ---------------------------------------------------------------------- */

/* Raw (name,value) pair as parsed from the query string (reference src/vector_index.c:210-211). */
typedef struct {
  const char *name;
  size_t nameLen;
  const char *value;
  size_t valLen;
} VecSimRawParam;

typedef struct {
  size_t efRuntime;
  double epsilon;
} HNSWRuntimeParams;

typedef struct {
  size_t windowSize;
  size_t bufferCapacity;
  VecSimOptionMode searchHistory;
  double epsilon;
} SVSRuntimeParams;

typedef struct {
  VecSimBool shouldRerank; /* reference src/iterators/hybrid_reader.c:249 */
} HNSWDiskRuntimeParams;

/* Members used: reference src/iterators/hybrid_reader.c:351,404,656; src/vector_index.c:268;
 * src/redisearch_rs/vector_score_source/src/test_utils.rs:219-220.  FLAT reads only batchSize,
 * searchMode and timeoutCtx. */
typedef struct {
  union {
    HNSWRuntimeParams hnswRuntimeParams;
    SVSRuntimeParams svsRuntimeParams;
    HNSWDiskRuntimeParams hnswDiskRuntimeParams;
  };
  size_t batchSize;
  VecSearchMode searchMode;
  void *timeoutCtx;
} VecSimQueryParams;

/* ---- info ------------------------------------------------------------------------------------ */

/* .dim/.type/.metric reference src/vector_index.c:241-244; .algo/.isMulti
 * src/debug_commands.c:1814-1819; .isDisk src/redisearch_rs/c_wrappers/vecsim/src/index.rs:151. */
typedef struct {
  VecSimAlgo algo;
  VecSimMetric metric;
  VecSimType type;
  bool isMulti;
  bool isTiered;
  bool isDisk;
  size_t blockSize;
  size_t dim;
} VecSimIndexBasicInfo;

/* reference src/info/field_spec_info.c:281-285 */
typedef struct {
  size_t memory;
  size_t numberOfMarkedDeleted;
  size_t directHNSWInsertions;
  size_t flatBufferSize;
} VecSimIndexStatsInfo;

/* Debug-info iterator types live in VecSim/info_iterator.h (reference src/debug_commands.c:49). */
typedef struct VecSimDebugInfoIterator VecSimDebugInfoIterator;

/* ---- process-wide hooks (installed at reference src/module-init/module-init.c:147-151) -------- */

typedef void *(*allocFn)(size_t n);
typedef void *(*callocFn)(size_t nelem, size_t elemsz);
typedef void *(*reallocFn)(void *p, size_t n);
typedef void (*freeFn)(void *p);

typedef struct {
  allocFn allocFunction;
  callocFn callocFunction;
  reallocFn reallocFunction;
  freeFn freeFunction;
} VecSimMemoryFunctions;

/* returns non-zero when the query owning `ctx` has run out of time */
typedef int (*timeoutCallbackFunction)(void *ctx);
/* reference src/vector_index.c:803 VecSimLogCallback */
typedef void (*logCallbackFunction)(void *ctx, const char *level, const char *message);

#define VecSimCommonStrings_LOG_VERBOSE_STRING "verbose"
#define VecSimCommonStrings_LOG_NOTICE_STRING "notice"
#define VecSimCommonStrings_LOG_WARNING_STRING "warning"
#define VecSimCommonStrings_LOG_DEBUG_STRING "debug"

#ifdef __cplusplus
}
#endif
#endif /* VECSIM_VEC_SIM_COMMON_H */
