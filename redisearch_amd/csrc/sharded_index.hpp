// sharded_index.hpp -- the engine side of one FLAT index over several GPUs of one process (sharded_index.cpp).
// Two doors lead here: the explicit extension API RSGPU_ShardedIndex_* (include/rsgpu_ext.h), and -- with the
// "shards" knob set -- VecSimIndex_New itself, whose handle then serves the WHOLE VecSim C ABI (queries, batch
// iterator, ad-hoc context, writes, info) from the shards: multi-GPU behind the reference's unchanged boundary.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "flat_index.hpp"

struct RSGPU_ShardedIndex;
struct RSGPU_ShardComm;

namespace rsgpu {

// the exchange over RCCL (shard_comm.cpp): one communicator per device for ranks that live in one process, and the
// exchange itself -- this rank's winners -> ncclAllGather -> merge kernel -> the global k best
std::vector<RSGPU_ShardComm *> shard_comm_init_group(const std::vector<int> &devices);  // all ranks, from the calling thread
size_t shard_comm_exchange_group(const std::vector<RSGPU_ShardComm *> &cs, const VecSimQueryResult *const *local, const size_t *n_local,
                                 size_t k, uint64_t *labels_out, double *scores_out);
void shard_comm_unique_id(void *id128);  // 128 bytes
RSGPU_ShardComm *shard_comm_init_rank(int rank, int world, const void *id128, int device);  // on the rank's own thread
size_t shard_comm_exchange(RSGPU_ShardComm *c, const VecSimQueryResult *local, size_t n_local, size_t k, uint64_t *labels_out,
                           double *scores_out);

RSGPU_ShardedIndex *sharded_new(const BFParams &p, void *log_ctx, int n_shards, const int *devices, bool replicas);
void sharded_free(RSGPU_ShardedIndex *si);
void *sharded_log_ctx(RSGPU_ShardedIndex *si);
FlatIndex *sharded_first(RSGPU_ShardedIndex *si);  // shard 0: type / metric / dim of the index
int sharded_add(RSGPU_ShardedIndex *si, const void *blob, size_t label);
int sharded_remove(RSGPU_ShardedIndex *si, size_t label);
size_t sharded_size(RSGPU_ShardedIndex *si);
size_t sharded_label_count(RSGPU_ShardedIndex *si);
size_t sharded_memory(RSGPU_ShardedIndex *si);
int sharded_last_mode(RSGPU_ShardedIndex *si);
VecSimQueryReply *sharded_topk(RSGPU_ShardedIndex *si, const void *query, size_t k, VecSimQueryParams *qp,
                               VecSimQueryReply_Order order);
VecSimQueryReply *sharded_range(RSGPU_ShardedIndex *si, const void *query, double radius, VecSimQueryParams *qp,
                                VecSimQueryReply_Order order);
double sharded_distance_from(RSGPU_ShardedIndex *si, size_t label, const void *normalized_blob);
bool sharded_prefer_adhoc(RSGPU_ShardedIndex *si, size_t subset, size_t k, bool initial_check);
void sharded_reserve(RSGPU_ShardedIndex *si, size_t rows);
// n synthetic rows split in contiguous runs over the shards (replicas: every shard gets all of them)
long sharded_add_philox_rows(RSGPU_ShardedIndex *si, uint64_t seed, uint64_t first_index, size_t n, size_t first_label);

// batch iterator over all shards: per-shard iterators + look-ahead buffers merged by (score, label)
struct ShardedBatchIterator;
ShardedBatchIterator *sharded_batch_new(RSGPU_ShardedIndex *si, const void *query, VecSimQueryParams *qp);
bool sharded_batch_has_next(ShardedBatchIterator *it);
VecSimQueryReply *sharded_batch_next(ShardedBatchIterator *it, size_t n, VecSimQueryReply_Order order);
void sharded_batch_reset(ShardedBatchIterator *it);
void sharded_batch_free(ShardedBatchIterator *it);

// ad-hoc brute-force context: the labels of one call are routed to the shards that own them
struct ShardedAdhoc;
ShardedAdhoc *sharded_adhoc_new(RSGPU_ShardedIndex *si, const void *query);
void sharded_adhoc_distances(ShardedAdhoc *a, const size_t *labels, double *out, size_t count);
void sharded_adhoc_free(ShardedAdhoc *a);

}  // namespace rsgpu
