// flat_index.cpp -- host drivers of the MI355X FLAT index (see flat_index.hpp for the layout).
//
// Restates, as GPU orchestration, what the reference gets from VecSim's BruteForceIndex through
// VecSimIndex_{AddVector,DeleteVector,TopKQuery,RangeQuery,GetDistanceFrom_Unsafe,PreferAdHocSearch}
// (call sites: reference src/document.c:721, src/indexer.c:186, src/iterators/hybrid_reader.c:316,374,
// src/vector_index.c:152).  No CPU distance code exists here: without a device every entry point
// fails loudly.
#include "flat_index.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>

namespace rsgpu {

// ---------------------------------------------------------------------------------------------------
Hooks &hooks() {
  static Hooks h;
  return h;
}

void logf(void *ctx, const char *level, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (hooks().log) hooks().log(ctx, level, buf);
  else if (getenv("RSGPU_VERBOSE")) fprintf(stderr, "[rsgpu:%s] %s\n", level, buf);
}

bool device_available(std::string *why) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    if (why) *why = std::string("no HIP device: ") + hipGetErrorString(e);
    return false;
  }
  return true;
}

bool timed_out(void *timeout_ctx) {
  timeoutCallbackFunction cb = hooks().timeout;
  return cb && cb(timeout_ctx) != 0;
}

std::string &last_error() {
  static thread_local std::string e;
  return e;
}

ScanProfile &scan_profile() {
  static ScanProfile p;
  return p;
}

CoalesceStats &coalesce_stats() {
  static CoalesceStats t;
  return t;
}
TwoStageStats &two_stage_stats() {
  static TwoStageStats t;
  return t;
}

VecSimQueryReply *new_reply(size_t len, VecSimQueryReply_Code code) {
  VecSimQueryReply *r = host_alloc<VecSimQueryReply>(1);
  r->results = len ? host_alloc<VecSimQueryResult>(len) : nullptr;
  r->len = len;
  r->code = code;
  return r;
}

// ---- fp16 / bf16 on the host (blob normalisation only) ---------------------------------------------
static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (!man) bits = sign;
    else {
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp + 112) << 23) | (man << 13);
  float f;
  memcpy(&f, &bits, 4);
  return f;
}
static uint16_t f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, a = x & 0x7fffffffu;
  if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0));
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (a < 0x33000001u) return (uint16_t)sign;
  int e = (int)(a >> 23) - 127;
  uint32_t man = (a & 0x7fffffu) | 0x800000u;
  int shift = e < -14 ? 13 + (-14 - e) : 13;
  uint32_t half = man >> shift, rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
  if (rem > mid || (rem == mid && (half & 1))) half++;
  if (e < -14) return (uint16_t)(sign | half);
  return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (half - 0x400u)));
}
static float bf2f(uint16_t h) {
  uint32_t b = (uint32_t)h << 16;
  float f;
  memcpy(&f, &b, 4);
  return f;
}
static uint16_t f2bf(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
  x += 0x7fffu + ((x >> 16) & 1);
  return (uint16_t)(x >> 16);
}

// VecSim_Normalize semantics (reference src/iterators/hybrid_reader.c:304): in place, fp32 math.
void normalize_blob(void *blob, size_t dim, VecSimType type) {
  switch (type) {
    case VecSimType_FLOAT32: {
      float *v = (float *)blob;
      float s[8] = {0};
      for (size_t i = 0; i < dim; i++) s[i & 7] += v[i] * v[i];
      float n = sqrtf(((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7])));
      for (size_t i = 0; i < dim; i++) v[i] /= n;
      break;
    }
    case VecSimType_FLOAT64: {
      double *v = (double *)blob, s = 0;
      for (size_t i = 0; i < dim; i++) s += v[i] * v[i];
      s = sqrt(s);
      for (size_t i = 0; i < dim; i++) v[i] /= s;
      break;
    }
    case VecSimType_FLOAT16:
    case VecSimType_BFLOAT16: {
      uint16_t *v = (uint16_t *)blob;
      bool half = type == VecSimType_FLOAT16;
      float s = 0;
      for (size_t i = 0; i < dim; i++) {
        float a = half ? h2f(v[i]) : bf2f(v[i]);
        s += a * a;
      }
      float n = sqrtf(s);
      for (size_t i = 0; i < dim; i++) {
        float a = (half ? h2f(v[i]) : bf2f(v[i])) / n;
        v[i] = half ? f2h(a) : f2bf(a);
      }
      break;
    }
    case VecSimType_INT8:
    case VecSimType_UINT8: {  // elements untouched, fp32 norm appended behind them
      long long s = 0;
      for (size_t i = 0; i < dim; i++) {
        int a = type == VecSimType_INT8 ? (int)((int8_t *)blob)[i] : (int)((uint8_t *)blob)[i];
        s += (long long)a * a;
      }
      float n = sqrtf((float)s);
      memcpy((char *)blob + dim, &n, 4);
      break;
    }
    default: break;
  }
}

// ---- workspaces -------------------------------------------------------------------------------------
template <typename T>
static void dev_realloc(T *&p, size_t old_n, size_t new_n) {
  if (p) {
    HIP_CHECK(hipFree(p));
    CtxPool::get().account(-(long)(old_n * sizeof(T)));
  }
  p = nullptr;
  HIP_CHECK(hipMalloc((void **)&p, new_n * sizeof(T)));
  CtxPool::get().account((long)(new_n * sizeof(T)));
}
template <typename T>
static void pin_realloc(T *&p, size_t new_n) {
  if (p) HIP_CHECK(hipHostFree(p));
  p = nullptr;
  HIP_CHECK(hipHostMalloc((void **)&p, new_n * sizeof(T), hipHostMallocDefault));
}

QueryCtx::QueryCtx(int dev) : device(dev) {
  HIP_CHECK(hipSetDevice(dev));
  HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreate(&ev0));
  HIP_CHECK(hipEventCreate(&ev1));
  dev_realloc(d_hist, 0, kSelectLevelsMax * 256);
  dev_realloc(d_counters, 0, 4);
  dev_realloc(d_bound, 0, 2);
  dev_realloc(d_tau, 0, 1);
  dev_realloc(d_cand, 0, kCandCap);
  dev_realloc(d_fcnt, 0, 4);
  pin_realloc(h_fcnt, 4);
  pin_realloc(h_counters, 16);
  dev_realloc(d_mq_tau, 0, kMqMaxQueries);
  dev_realloc(d_mq_cnt, 0, kMqMaxQueries);
  pin_realloc(h_mq_n, 2 * kMqMaxQueries);
  h_mq_over = h_mq_n + kMqMaxQueries;
}
#define HIP_IGNORE(x) (void)(x)
QueryCtx::~QueryCtx() {
  HIP_IGNORE(hipSetDevice(device));
  HIP_IGNORE(hipStreamSynchronize(stream));
  void *dev[] = {d_query, d_keys, d_hist, d_counters, d_bound, d_out_rows, d_out_keys, d_ids, d_dists, d_tau, d_cand, d_fcnt,
                 d_mq_queries, d_mq_tau, d_mq_cnt, d_mq_cand};
  if (h_fcnt) HIP_IGNORE(hipHostFree(h_fcnt));
  void *pin[] = {h_query, h_out_rows, h_out_keys, h_counters, h_ids, h_dists, h_mq_queries, h_mq_n};
  for (void *p : dev) if (p) HIP_IGNORE(hipFree(p));
  for (void *p : pin) if (p) HIP_IGNORE(hipHostFree(p));
  HIP_IGNORE(hipEventDestroy(ev0));
  HIP_IGNORE(hipEventDestroy(ev1));
  HIP_IGNORE(hipStreamDestroy(stream));
}
void QueryCtx::ensure_query(size_t bytes) {
  if (bytes <= query_cap) return;
  size_t cap = round_up(bytes, 4096);
  dev_realloc(d_query, query_cap, cap);
  pin_realloc(h_query, cap);
  query_cap = cap;
  cached_query_owner = 0;
}
void QueryCtx::ensure_mq(size_t query_bytes) {
  if (!d_mq_cand) dev_realloc(d_mq_cand, 0, (size_t)kMqMaxQueries * kCandCap);
  if (query_bytes <= mq_query_cap) return;
  const size_t cap = round_up(query_bytes, 4096);
  dev_realloc(d_mq_queries, mq_query_cap, cap);
  pin_realloc(h_mq_queries, cap);
  mq_query_cap = cap;
}
void QueryCtx::ensure_keys(size_t rows) {
  if (rows <= keys_cap) return;
  size_t cap = round_up(rows + rows / 8 + 1024, 1024);
  dev_realloc(d_keys, keys_cap, cap);
  keys_cap = cap;
}
void QueryCtx::ensure_out(size_t k) {
  if (k <= out_cap) return;
  size_t cap = round_up(k + k / 2 + 256, 256);
  dev_realloc(d_out_rows, out_cap, cap);
  dev_realloc(d_out_keys, out_cap, cap);
  pin_realloc(h_out_rows, cap);
  pin_realloc(h_out_keys, cap);
  out_cap = cap;
}
void QueryCtx::ensure_gather(size_t m) {
  if (m <= gather_cap) return;
  size_t cap = round_up(m + m / 2 + 256, 256);
  dev_realloc(d_ids, gather_cap, cap);
  dev_realloc(d_dists, 2 * gather_cap, 2 * cap);  // fp32 distances, or fp64 for FLOAT64 indexes
  pin_realloc(h_ids, cap);
  pin_realloc(h_dists, 2 * cap);
  gather_cap = cap;
}

CtxPool &CtxPool::get() {
  static CtxPool *p = new CtxPool();  // intentionally leaked: outlives static destructors
  return *p;
}
QueryCtx *CtxPool::acquire(int device) {
  {
    std::lock_guard<std::mutex> g(mu_);
    for (size_t i = 0; i < idle_.size(); i++)
      if (idle_[i]->device == device) {
        QueryCtx *c = idle_[i];
        idle_.erase(idle_.begin() + (long)i);
        return c;
      }
  }
  return new QueryCtx(device);
}
void CtxPool::release(QueryCtx *c) {
  std::lock_guard<std::mutex> g(mu_);
  idle_.push_back(c);
}
void CtxPool::drain() {
  std::vector<QueryCtx *> v;
  {
    std::lock_guard<std::mutex> g(mu_);
    v.swap(idle_);
  }
  for (QueryCtx *c : v) delete c;
}

// ---- index -----------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_uid{1};

FlatIndex::FlatIndex(const BFParams &p, void *lctx)
    : type(p.type), metric(p.metric), dim(p.dim), multi(p.multi),
      block_size(p.blockSize ? p.blockSize : DEFAULT_BLOCK_SIZE), log_ctx(lctx),
      labels_(p.multi, round_up(p.dim * type_size(p.type), 16), &row_label_, &host_bytes_) {
  ktype = (int)type;
  const bool int_type = type == VecSimType_INT8 || type == VecSimType_UINT8;
  kmetric = metric == VecSimMetric_L2 ? KM_L2 : (metric == VecSimMetric_Cosine && int_type) ? KM_COS : KM_IP;
  key_bytes = key_bytes_of(ktype);
  elem_bytes_ = dim * type_size(type);
  stride_ = round_up(elem_bytes_, 16);
  if (type == VecSimType_FLOAT32 && !multi) {
    if (scan_tuning().shadow8) shadow_ = 2;  // every metric (the error band carries the norms)
    else if (scan_tuning().shadow16 && metric == VecSimMetric_Cosine) shadow_ = 1;
  }
  // FLOAT16 IP / cosine: an int8 shadow with one index-wide scale for the batched MFMA pass (batch_query.cpp)
  // (BFLOAT16 the same way since round 3: tests/test_gpu_batch_i8_shadow.py)
  if ((type == VecSimType_FLOAT16 || type == VecSimType_BFLOAT16) && !multi && metric != VecSimMetric_L2 && scan_tuning().shadow8)
    shadow_ = 3;
  // ... and WITHOUT the knob (round 6): the same int8 passes over the fp16 rows themselves, quantised in flight -- nothing stored
  if (type == VecSimType_FLOAT16 && !multi && metric != VecSimMetric_L2 && !shadow_ && scan_tuning().gemm_qs_h8 &&
      gemm_qs_h8_supported((uint32_t)(stride_ / 16)))
    h8_ = true;
  // FLOAT32 IP / cosine (knob gemm_qs_f8): the same, through gemm_qs_h8r_kernel<.., SRC_F8>
  if (type == VecSimType_FLOAT32 && !multi && metric != VecSimMetric_L2 && !shadow_ && scan_tuning().gemm_qs_f8 &&
      gemm_qs_f8_supported((uint32_t)(stride_ / 16)))
    h8_ = true;
  sstride_ = shadow_ == 1 ? round_up(dim * 2, 16) : (shadow_ >= 2 ? round_up(dim, 16) : 0);
  uid = g_uid++;
  HIP_CHECK(hipGetDevice(&device));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  scan_tuning().num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_CHECK(hipStreamCreateWithFlags(&wstream_, hipStreamNonBlocking));
  if (shadow_ == 2) {
    HIP_CHECK(hipMalloc((void **)&d_smax_, 4 * sizeof(uint32_t)));
    HIP_CHECK(hipMemset(d_smax_, 0, 4 * sizeof(uint32_t)));
  }
  if (shadow_ == 3 || h8_ || (shadow_ == 2 && metric != VecSimMetric_L2)) {  // (FLOAT16, or FLOAT32 next to its per-row shadow)
    HIP_CHECK(hipMalloc((void **)&d_s8g_stats_, 4 * sizeof(uint32_t)));
    HIP_CHECK(hipMemset(d_s8g_stats_, 0, 4 * sizeof(uint32_t)));
  }
  // staging block: up to 4096 rows or 8 MiB
  stage_cap_ = std::max<size_t>(1, std::min<size_t>(4096, (8u << 20) / stride_));
  HIP_CHECK(hipHostMalloc((void **)&h_stage_, stage_cap_ * stride_, hipHostMallocDefault));
  size_t init = block_size;
  if (p.initialCapacity && p.initialCapacity != SIZE_MAX) init = std::max(init, std::min<size_t>(p.initialCapacity, 1u << 20));
  grow(init);
}

FlatIndex::~FlatIndex() {
  HIP_IGNORE(hipSetDevice(device));
  HIP_IGNORE(hipStreamSynchronize(wstream_));
  rows_buf_.release();
  shadow_buf_.release();
  if (d_labels_) HIP_IGNORE(hipFree(d_labels_));
  if (d_sscale_) HIP_IGNORE(hipFree(d_sscale_));
  if (d_smax_) HIP_IGNORE(hipFree(d_smax_));
  if (d_s8g_stats_) HIP_IGNORE(hipFree(d_s8g_stats_));
  if (d_s8g_f32_) HIP_IGNORE(hipFree(d_s8g_f32_));
  if (d_hnorm_) HIP_IGNORE(hipFree(d_hnorm_));
  if (d_hn_bad_) HIP_IGNORE(hipFree(d_hn_bad_));
  if (h_stage_) HIP_IGNORE(hipHostFree(h_stage_));
  HIP_IGNORE(hipStreamDestroy(wstream_));
}

size_t FlatIndex::memory() const {
  return rows_buf_.physical() + shadow_buf_.physical() + s8g_f32_cap_rows_ * round_up(dim, 16) + hnorm_cap_rows_ * sizeof(float) +
         cap_rows_ * ((shadow_ == 2 ? 8 : 0) + sizeof(uint64_t)) +
         host_bytes_ + labels_.device_bytes() + stage_cap_ * stride_;
}

void FlatIndex::grow(size_t min_rows) {
  // 32 rows of slack behind the rows: the batched filter pass reads its ragged last tile whole
  const size_t need_row_bytes = (min_rows + 32) * stride_;
  // (the shadow is its own buffer with its own rounding: a mapped row matrix rounds up to a 256 MiB / 1 GiB chunk and
  // covers far more rows than a 2-4x smaller shadow that is still a plain allocation of exactly what was asked for)
  const bool shadow_fits = !shadow_ || (min_rows + 32) * sstride_ <= shadow_buf_.capacity();
  if (min_rows <= cap_rows_ && need_row_bytes <= rows_buf_.capacity() && shadow_fits) return;
  if (min_rows > 0xFFFFFFF0ull) throw std::runtime_error("FLAT index is limited to 2^32 rows per device");
  const int mode = scan_tuning().vmm;
  // Row matrix (+ shadow): no copy once mapped (grow_buffer.hpp), so it grows by what is needed (rounded to a
  // physical chunk by the buffer); while it is a plain allocation it doubles like the label array below.
  size_t next = cap_rows_ < (1u << 22) ? cap_rows_ * 2 : cap_rows_ + cap_rows_ / 2;
  size_t new_cap = round_up(std::max(min_rows, next), 64);
  // a buffer that is (or is about to be) mapped grows by what is needed; a plain allocation -- small buffers, VMM switched
  // off, or a mapping the driver refused (GrowBuffer::vmm_broken) -- is copied on every growth and therefore doubles
  auto target_rows = [&](const GrowBuffer &b, size_t row_bytes) {
    const bool will_map = mode == 1 && !b.vmm_broken && (b.mapped() || ((min_rows + 32) * row_bytes >= GrowBuffer::kVmmThreshold && vmm_supported(device)));
    return will_map ? min_rows : new_cap;
  };
  rows_buf_.reserve_factor = shadow_buf_.reserve_factor = scan_tuning().vmm_reserve_factor;
  rows_buf_.chunk_override = shadow_buf_.chunk_override = (size_t)scan_tuning().vmm_chunk_mib << 20;
  rows_buf_.ensure(device, (target_rows(rows_buf_, stride_) + 32) * stride_, (size_t)n_rows_ * stride_, wstream_, mode);
  d_rows_ = rows_buf_.ptr();
  if (shadow_) {
    shadow_buf_.ensure(device, (target_rows(shadow_buf_, sstride_) + 32) * sstride_, (size_t)n_rows_ * sstride_, wstream_, mode);
    d_shadow_ = shadow_buf_.ptr();
  }
  if (min_rows <= cap_rows_) return;
  // labels (8 B per row) and int8 row scales (4 B): small next to the rows -- allocate, copy, swap
  uint64_t *nl = nullptr;
  float *nsc = nullptr;
  struct Rollback {
    void **p[2];
    bool armed = true;
    ~Rollback() {
      if (armed)
        for (void **q : p)
          if (*q) HIP_IGNORE(hipFree(*q));
    }
  } rollback{{(void **)&nl, (void **)&nsc}};
  HIP_CHECK(hipMalloc((void **)&nl, new_cap * sizeof(uint64_t)));
  if (shadow_ == 2) HIP_CHECK(hipMalloc((void **)&nsc, new_cap * 2 * sizeof(float)));
  if (n_rows_) {
    HIP_CHECK(hipMemcpyAsync(nl, d_labels_, (size_t)n_rows_ * sizeof(uint64_t), hipMemcpyDeviceToDevice, wstream_));
    if (nsc) HIP_CHECK(hipMemcpyAsync(nsc, d_sscale_, (size_t)n_rows_ * 2 * sizeof(float), hipMemcpyDeviceToDevice, wstream_));
    HIP_CHECK(hipStreamSynchronize(wstream_));
  }
  rollback.armed = false;
  if (d_labels_) HIP_CHECK(hipFree(d_labels_));
  d_labels_ = nl;
  if (shadow_ == 2) {
    if (d_sscale_) HIP_CHECK(hipFree(d_sscale_));
    d_sscale_ = nsc;
  }
  cap_rows_ = new_cap;
  labels_.set_row_capacity(new_cap);
}

// new rows [row_begin,row_end) -> shadow, queued on wstream_ behind the copies / normalisation that produced them;
// the int8 form also refreshes the host copy of the largest row scale (the caller syncs wstream_ anyway)
void FlatIndex::shadow_convert(uint32_t row_begin, uint32_t row_end) {
  if (!shadow_ || shadow_ == 3 || row_end <= row_begin) return;  // (3: built lazily, ensure_shadow8g)
  if (shadow_ == 1) {
    launch_shadow_rows(d_rows_, stride_, (uint32_t)dim, row_begin, row_end, d_shadow_, sstride_, wstream_);
    return;
  }
  launch_shadow8_rows(d_rows_, stride_, (uint32_t)dim, row_begin, row_end, d_shadow_, sstride_, d_sscale_, d_smax_, wstream_);
  uint32_t bits[4] = {0, 0, 0, 0};
  HIP_CHECK(hipMemcpyAsync(bits, d_smax_, sizeof bits, hipMemcpyDeviceToHost, wstream_));
  HIP_CHECK(hipStreamSynchronize(wstream_));
  memcpy(&s_max_, &bits[0], 4);
  memcpy(&n2_max_, &bits[1], 4);
  s_bad_ = bits[2] != 0;
}

void FlatIndex::reserve(size_t rows) {
  std::unique_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  grow(rows);
}

// cosine rows/queries are stored normalised -- except INT8/UINT8, whose elements cannot carry a norm
// (the scan divides by |x| and |q| instead, kernels.hpp KM_COS)
void FlatIndex::normalize_host(void *blob) const {
  if (type == VecSimType_INT8 || type == VecSimType_UINT8) return;
  normalize_blob(blob, dim, type);
}

// int8 shadow with one index-wide scale (shadow_ == 3), brought up to date with the rows: see flat_index.hpp.
bool FlatIndex::ensure_shadow8g() {
  if (!s8g_enabled()) return false;
  flush_if_needed();
  {
    std::shared_lock<std::shared_mutex> g(mu);
    if (s_bad_) return false;
    if (s8g_built_ >= n_rows_ && s8g_seen_ >= n_rows_ && s8g_scale_ > 0.0f) return true;
  }
  std::unique_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  const uint32_t n = n_rows_;
  if (!n) return false;
  s8g_built_ = std::min(s8g_built_, n);
  s8g_seen_ = std::min(s8g_seen_, n);
  if (shadow_ != 3 && !h8_ && (size_t)n + 32 > s8g_f32_cap_rows_) {  // FLOAT32: the int8 rows live in their own allocation
    if (d_s8g_f32_) HIP_IGNORE(hipFree(d_s8g_f32_));
    d_s8g_f32_ = nullptr;
    s8g_f32_cap_rows_ = 0;
    const size_t cap = (size_t)n + n / 8 + 64;
    HIP_CHECK(hipMalloc((void **)&d_s8g_f32_, cap * s8g_stride()));
    s8g_f32_cap_rows_ = cap;
    s8g_built_ = 0;
  }
  uint32_t st[4] = {0, 0, 0, 0};
  if (s8g_seen_ < n) {  // the largest |x_i| of the rows not looked at yet
    launch_absmax_rows(ktype, d_rows_, stride_, (uint32_t)dim, s8g_seen_, n, d_s8g_stats_, wstream_);
    s8g_seen_ = n;
  }
  HIP_CHECK(hipMemcpyAsync(st, d_s8g_stats_, sizeof st, hipMemcpyDeviceToHost, wstream_));
  HIP_CHECK(hipStreamSynchronize(wstream_));
  float gmax;
  memcpy(&gmax, &st[0], 4);
  if (st[3] || !(gmax <= 3.0e38f)) {  // inf / NaN element: no scale bounds such an index
    s_bad_ = true;
    return false;
  }
  float want = gmax > 0.0f ? gmax / 127.0f : 1.0f;
  uint16_t inv_bits = 0;
  float f8_inv = 0.0f;
  if (h8_ && type == VecSimType_FLOAT32) {  // ... inv = the largest fp32 <= 127 / max |x_i|
    f8_inv = gmax > 0.0f ? std::min(127.0f / gmax, 3.0e38f) : 127.0f;
    while (gmax > 0.0f && (double)f8_inv * (double)gmax > 127.0) f8_inv = std::nextafterf(f8_inv, 0.0f);
    want = 1.0f / f8_inv;
  } else if (h8_) {  // the scale of the in-flight quantiser is 1 / inv, inv = the largest fp16 <= 127 / max |x_i| (h8_quant.hpp)
    const float target = gmax > 0.0f ? std::min(127.0f / gmax, 65504.0f) : 127.0f;
    _Float16 inv = (_Float16)target;
    memcpy(&inv_bits, &inv, 2);
    if ((float)inv > target) {  // rounded up: one fp16 step down (positive finite: the bit pattern orders like the value)
      inv_bits--;
      memcpy(&inv, &inv_bits, 2);
    }
    want = 1.0f / (float)inv;
  }
  if (!(s8g_scale_ > 0.0f) || want > s8g_scale_) {  // first build, or a row outgrew the scale: every row again
    s8g_scale_ = want;
    h8_inv_bits_ = inv_bits;
    f8_inv_ = f8_inv;
    s8g_built_ = 0;
    const uint32_t zero2[2] = {0, 0};  // the error maxima belong to the scale
    HIP_CHECK(hipMemcpyAsync(d_s8g_stats_ + 1, zero2, sizeof zero2, hipMemcpyHostToDevice, wstream_));
  }
  if (s8g_built_ < n && h8_) {  // nothing to store: the maxima of the rows not covered yet
    if (type == VecSimType_FLOAT32) launch_f8_stats(d_rows_, stride_, (uint32_t)dim, s8g_built_, n, f8_inv_, d_s8g_stats_, wstream_);
    else launch_h8_stats(d_rows_, stride_, (uint32_t)dim, s8g_built_, n, h8_inv_bits_, d_s8g_stats_, wstream_);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(wstream_));
    s8g_built_ = n;
  }
  if (s8g_built_ < n) {
    launch_shadow8g_rows(ktype, d_rows_, stride_, (uint32_t)dim, s8g_built_, n, s8g_scale_, const_cast<uint8_t *>(s8g_rows()),
                         s8g_stride(), d_s8g_stats_, wstream_);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(wstream_));
    s8g_built_ = n;
  }
  return true;
}

// |x|^2 / 2 per row for the L2 form of the batched matrix-core pass: see flat_index.hpp.
bool FlatIndex::ensure_half_norms() {
  flush_if_needed();
  {
    std::shared_lock<std::shared_mutex> g(mu);
    if (hn_bad_) return false;
    if (d_hnorm_ && hn_built_ >= n_rows_ && n_rows_) return true;
  }
  std::unique_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  const uint32_t n = n_rows_;
  if (!n || hn_bad_) return false;
  hn_built_ = std::min(hn_built_, n);
  if (!d_hn_bad_) {
    HIP_CHECK(hipMalloc((void **)&d_hn_bad_, 2 * sizeof(uint32_t)));
    HIP_CHECK(hipMemset(d_hn_bad_, 0, 2 * sizeof(uint32_t)));
  }
  if ((size_t)n + 96 > hnorm_cap_rows_) {  // (the pass reads 64 norms from the first row of its last tile on)
    if (d_hnorm_) HIP_IGNORE(hipFree(d_hnorm_));
    d_hnorm_ = nullptr;
    hnorm_cap_rows_ = 0;
    const size_t cap = (size_t)n + n / 8 + 128;
    HIP_CHECK(hipMalloc((void **)&d_hnorm_, cap * sizeof(float)));
    HIP_CHECK(hipMemsetAsync(d_hnorm_, 0, cap * sizeof(float), wstream_));
    hnorm_cap_rows_ = cap;
    hn_built_ = 0;
  }
  if (hn_built_ < n) {
    launch_half_norm_rows(ktype, d_rows_, stride_, hn_built_, n, 1.0f - 0.5f * hn_rel(), d_hnorm_, d_hn_bad_, wstream_);
    launch_max_f32_bits(d_hnorm_, hn_built_, n, d_hn_bad_ + 1, wstream_);
    HIP_CHECK(hipGetLastError());
    uint32_t bad[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(bad, d_hn_bad_, sizeof bad, hipMemcpyDeviceToHost, wstream_));
    HIP_CHECK(hipStreamSynchronize(wstream_));
    if (bad[0]) {  // a row with an inf / NaN norm: no band bounds such an index
      hn_bad_ = true;
      return false;
    }
    float stored_max;
    memcpy(&stored_max, &bad[1], 4);
    hn_max_ = stored_max / (1.0f - 0.5f * hn_rel()) * 1.000001f;  // (the stored norms are shrunk)
    hn_built_ = n;
  }
  return true;
}

float FlatIndex::half_sq_norm_host(const void *blob) const {
  float s = 0.0f;
  switch (type) {
    case VecSimType_FLOAT32:
      for (size_t i = 0; i < dim; i++) s += ((const float *)blob)[i] * ((const float *)blob)[i];
      break;
    case VecSimType_FLOAT16:
      for (size_t i = 0; i < dim; i++) {
        const float a = h2f(((const uint16_t *)blob)[i]);
        s += a * a;
      }
      break;
    case VecSimType_BFLOAT16:
      for (size_t i = 0; i < dim; i++) {
        const float a = bf2f(((const uint16_t *)blob)[i]);
        s += a * a;
      }
      break;
    default: return __builtin_nanf("");
  }
  return 0.5f * s;
}

void FlatIndex::flush_if_needed() {
  {
    std::shared_lock<std::shared_mutex> g(mu);
    if (!stage_n_) return;
  }
  std::unique_lock<std::shared_mutex> g(mu);
  flush();
}

void FlatIndex::flush() {
  if (!stage_n_) return;
  HIP_CHECK(hipSetDevice(device));
  grow((size_t)n_rows_ + stage_n_);
  HIP_CHECK(hipMemcpyAsync(d_rows_ + (size_t)n_rows_ * stride_, h_stage_, stage_n_ * stride_, hipMemcpyHostToDevice, wstream_));
  HIP_CHECK(hipMemcpyAsync(d_labels_ + n_rows_, row_label_.data() + n_rows_, stage_n_ * sizeof(uint64_t),
                           hipMemcpyHostToDevice, wstream_));
  shadow_convert(n_rows_, (uint32_t)(n_rows_ + stage_n_));
  labels_.sync_device(wstream_);  // the staged rows' table entries
  HIP_CHECK(hipStreamSynchronize(wstream_));
  n_rows_ += (uint32_t)stage_n_;
  stage_n_ = 0;
}

int FlatIndex::add(const void *blob, size_t label) {
  std::unique_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  int ret = 1;
  if (!multi) {
    std::vector<uint32_t> rows;
    rows_of(label, rows);
    if (!rows.empty()) {  // overwrite: drop the old vector first (SURVEY.md 8c viii)
      g.unlock();
      remove(label);
      g.lock();
      ret = 0;
    }
  }
  if (stage_n_ == stage_cap_) flush();
  uint8_t *dst = h_stage_ + stage_n_ * stride_;
  memset(dst, 0, stride_);
  memcpy(dst, blob, elem_bytes_);
  if (metric == VecSimMetric_Cosine) normalize_host(dst);
  uint32_t row = (uint32_t)(n_rows_ + stage_n_);
  labels_.insert(label, row, wstream_);
  row_label_.push_back(label);
  stage_n_++;
  return ret;
}

int FlatIndex::remove(size_t label) {
  std::unique_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  flush();
  std::vector<uint32_t> rows;
  rows_of(label, rows);
  if (rows.empty()) return 0;
  labels_.leave_identity(wstream_);
  std::sort(rows.begin(), rows.end(), std::greater<uint32_t>());
  for (uint32_t r : rows) {
    uint32_t last = n_rows_ - 1;
    if (r != last) {  // move the last row into the hole
      uint64_t moved = row_label_[last];
      HIP_CHECK(hipMemcpyAsync(d_rows_ + (size_t)r * stride_, d_rows_ + (size_t)last * stride_, stride_,
                               hipMemcpyDeviceToDevice, wstream_));
      HIP_CHECK(hipMemcpyAsync(d_labels_ + r, d_labels_ + last, sizeof(uint64_t), hipMemcpyDeviceToDevice, wstream_));
      if (s8g_enabled() && !h8_) s8g_built_ = std::min(s8g_built_, r);  // rows from r on are quantised again on demand
      // (h8_: nothing is stored per row and the maxima do not care where a row lies)
      hn_built_ = std::min(hn_built_, r);                        // ... and their half norms recomputed
      if (shadow_ && shadow_ != 3)
        HIP_CHECK(hipMemcpyAsync(d_shadow_ + (size_t)r * sstride_, d_shadow_ + (size_t)last * sstride_, sstride_,
                                 hipMemcpyDeviceToDevice, wstream_));
      if (shadow_ == 2)
        HIP_CHECK(hipMemcpyAsync(d_sscale_ + 2 * (size_t)r, d_sscale_ + 2 * (size_t)last, 2 * sizeof(float), hipMemcpyDeviceToDevice,
                                 wstream_));
      row_label_[r] = moved;
      labels_.move_row(moved, last, r);
    }
    row_label_.pop_back();
    n_rows_--;
    // (the row that is appended next takes slot n_rows_: nothing derived from the old occupant may count as built)
    s8g_built_ = std::min(s8g_built_, n_rows_);
    s8g_seen_ = std::min(s8g_seen_, n_rows_);
    hn_built_ = std::min(hn_built_, n_rows_);
  }
  labels_.erase_label(label);
  labels_.sync_device(wstream_);  // the two or three table entries this delete touched
  HIP_CHECK(hipStreamSynchronize(wstream_));
  layout_epoch++;
  return (int)rows.size();
}

// a single-value index holds one vector per label: a bulk load may not collide with a stored label
// (AddVector's overwrite semantics would need a row-by-row path; the caller uses AddVector for that)
void FlatIndex::check_bulk_labels(size_t n, size_t first_label) const {
  if (multi) return;
  if (labels_.any_in_range(first_label, n)) throw std::runtime_error("bulk load: label range overlaps stored labels");
}

void FlatIndex::commit_bulk_rows(size_t n, size_t first_label) {
  if (metric == VecSimMetric_Cosine && kmetric == KM_IP)
    launch_normalize_rows(d_rows_, stride_, (uint32_t)dim, ktype, n_rows_, (uint32_t)(n_rows_ + n), wstream_);
  shadow_convert(n_rows_, (uint32_t)(n_rows_ + n));
  const size_t old = row_label_.size();
  // still label == base + row for every row?  otherwise the table takes the new (label, row) pairs -- it appears now if this
  // very call ends identity labelling
  labels_.insert_range(first_label, (uint32_t)old, n, wstream_);
  row_label_.resize(old + n);
  for (size_t i = 0; i < n; i++) row_label_[old + i] = first_label + i;
  labels_.sync_device(wstream_);
  HIP_CHECK(hipMemcpyAsync(d_labels_ + n_rows_, row_label_.data() + old, n * sizeof(uint64_t), hipMemcpyHostToDevice, wstream_));
  HIP_CHECK(hipStreamSynchronize(wstream_));
  HIP_CHECK(hipGetLastError());
  n_rows_ += (uint32_t)n;
}

int FlatIndex::add_device_rows(const void *dev_rows, size_t n, size_t first_label) {
  std::unique_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  flush();
  if (!n) return 0;
  check_bulk_labels(n, first_label);
  grow((size_t)n_rows_ + n);
  uint8_t *dst = d_rows_ + (size_t)n_rows_ * stride_;
  if (stride_ == elem_bytes_) {
    HIP_CHECK(hipMemcpyAsync(dst, dev_rows, n * stride_, hipMemcpyDeviceToDevice, wstream_));
  } else {
    HIP_CHECK(hipMemsetAsync(dst, 0, n * stride_, wstream_));
    HIP_CHECK(hipMemcpy2DAsync(dst, stride_, dev_rows, elem_bytes_, elem_bytes_, n, hipMemcpyDeviceToDevice, wstream_));
  }
  commit_bulk_rows(n, first_label);
  return (int)n;
}

int FlatIndex::add_philox_rows(uint64_t seed, uint64_t first_index, size_t n, size_t first_label) {
  std::unique_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  flush();
  if (!n) return 0;
  check_bulk_labels(n, first_label);
  grow((size_t)n_rows_ + n);
  launch_philox_rows(d_rows_, stride_, (uint32_t)dim, ktype, seed, first_index, n_rows_, (uint32_t)n, wstream_);
  commit_bulk_rows(n, first_label);
  return (int)n;
}

void FlatIndex::read_rows(uint32_t row_begin, size_t n, void *host_out) {
  flush_if_needed();
  std::shared_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  if ((size_t)row_begin + n > n_rows_) throw std::runtime_error("read_rows: range beyond the stored rows");
  if (!n) return;
  HIP_CHECK(hipMemcpy2D(host_out, elem_bytes_, d_rows_ + (size_t)row_begin * stride_, stride_, elem_bytes_, n,
                        hipMemcpyDeviceToHost));
}

size_t FlatIndex::size() {
  std::shared_lock<std::shared_mutex> g(mu);
  return (size_t)n_rows_ + stage_n_;
}
bool FlatIndex::contains(size_t label) {
  std::shared_lock<std::shared_mutex> g(mu);
  return labels_.contains(label);
}
size_t FlatIndex::label_count() {
  std::shared_lock<std::shared_mutex> g(mu);
  return multi ? labels_.label_count() : (size_t)n_rows_ + stage_n_;
}

VecSimIndexBasicInfo FlatIndex::basic_info() const {
  VecSimIndexBasicInfo i;
  memset(&i, 0, sizeof i);
  i.algo = VecSimAlgo_BF;
  i.metric = metric;
  i.type = type;
  i.isMulti = multi;
  i.isTiered = false;
  i.isDisk = false;
  i.blockSize = block_size;
  i.dim = dim;
  return i;
}

// ---- query building blocks ---------------------------------------------------------------------------
void FlatIndex::upload_query(QueryCtx *c, const void *blob, bool normalize) {
  c->ensure_query(round_up(stride_ + 16, 16) + sstride_ + 16);  // (+ room for the shadow-typed copy of a two-stage scan)
  memset(c->h_query, 0, stride_);
  memcpy(c->h_query, blob, elem_bytes_);
  if (normalize && metric == VecSimMetric_Cosine) normalize_host(c->h_query);
  size_t bytes = stride_;
  if (type == VecSimType_INT8 || type == VecSimType_UINT8) {
    // one more chunk behind the padded query: {sum q^2 as i32/u32, |q| as f32, 0, 0}
    long long qq = 0;
    for (size_t i = 0; i < dim; i++) {
      int a = type == VecSimType_INT8 ? (int)((const int8_t *)blob)[i] : (int)((const uint8_t *)blob)[i];
      qq += (long long)a * a;
    }
    uint32_t extra[4] = {(uint32_t)qq, 0, 0, 0};
    float qn = sqrtf((float)qq);
    memcpy(&extra[1], &qn, 4);
    memcpy(c->h_query + stride_, extra, 16);
    bytes += 16;
  }
  HIP_CHECK(hipMemcpyAsync(c->d_query, c->h_query, bytes, hipMemcpyHostToDevice, c->stream));
  c->cached_query_owner = 0;
}

void FlatIndex::scan_all(QueryCtx *c, uint32_t n) {
  c->ensure_keys((size_t)n * (key_bytes / 4));
  const bool prof = scan_profile().enabled.load(std::memory_order_relaxed) != 0;
  // profiling brackets the scan launch with events on ITS stream; they are read back after the
  // query's own stream synchronisation (collect_profile), so the timed path gains no extra sync
  if (prof) HIP_CHECK(hipEventRecord(c->ev0, c->stream));
  launch_scan(d_rows_, stride_, (uint32_t)dim, ktype, kmetric, 0, n, c->d_query, c->d_keys, c->stream);
  HIP_CHECK(hipGetLastError());
  if (prof) {
    HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    c->prof_rows = n;
    c->prof_bytes_per_row = elem_bytes_;
    c->prof_pending = true;
  }
}

static void collect_profile(QueryCtx *c) {
  if (!c->prof_pending) return;
  c->prof_pending = false;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return;
  ScanProfile &pf = scan_profile();
  pf.launches++;
  pf.bytes += (uint64_t)c->prof_rows * c->prof_bytes_per_row;
  pf.nanos += (uint64_t)((double)ms * 1e6);
}

void radix_select(QueryCtx *c, const void *d_keys, int key_bytes, uint32_t n, uint32_t k, const Bound &lower,
                  std::vector<Hit> &out, Bound *upper) {
  out.clear();
  if (upper) *upper = lower;
  if (!k || !n) return;
  c->ensure_out(k);
  SelectBufs b{c->d_hist, c->d_counters, c->d_out_rows, c->d_out_keys, c->d_bound};
  HIP_CHECK(hipMemsetAsync(c->d_hist, 0, kSelectLevelsMax * 256 * sizeof(uint32_t), c->stream));
  HIP_CHECK(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(uint32_t), c->stream));
  const int levels = key_bytes + 4, has_lower = lower.valid ? 1 : 0;
  int done = 0;
  // key levels first; the four row levels only when equal keys straddle rank k
  for (int round = 0; round < 2; round++) {
    int upto = round == 0 ? key_bytes : levels;
    for (int p = done; p < upto; p++)
      launch_select_pass(d_keys, key_bytes, n, p, k, lower.key, lower.row, has_lower, b, c->stream);
    done = upto;
    launch_select_collect(d_keys, key_bytes, n, done, k, lower.key, lower.row, has_lower, b, (uint32_t)c->out_cap, c->stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(c->h_counters, c->d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(c->h_bound(), c->d_bound, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(c->h_out_rows, c->d_out_rows, k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(c->h_out_keys, c->d_out_keys, (size_t)k * key_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    collect_profile(c);
    if (c->h_counters[1] == 0) break;  // exact
    if (round == 1) throw std::runtime_error("radix select did not converge");
  }
  uint32_t got = std::min<uint32_t>(c->h_counters[0], k);
  out.resize(got);
  const uint32_t *k32 = reinterpret_cast<const uint32_t *>(c->h_out_keys);
  for (uint32_t i = 0; i < got; i++)
    out[i] = Hit{c->h_out_rows[i], key_bytes == 8 ? c->h_out_keys[i] : (uint64_t)k32[i]};
  std::sort(out.begin(), out.end(), [](const Hit &a, const Hit &b2) {
    return a.key != b2.key ? a.key < b2.key : a.row < b2.row;
  });
  if (upper) {
    upper->key = c->h_bound()[0];
    upper->row = (uint32_t)c->h_bound()[1];
    upper->valid = true;
  }
}

// Small-K fast path: K-th smallest group minimum of a spread key sample bounds the answer (tau), one streaming pass
// keeps the keys <= tau, a single workgroup selects the exact K among them (select_kernels.hip
// "threshold filter").  ~40 us after the scan instead of ~110 us for the four histogram levels.
static bool filter_select(QueryCtx *c, const uint32_t *d_keys, uint32_t n, uint32_t k, std::vector<Hit> &out) {
  // tau = k-th smallest of 1024 group minima over a 64 Ki key sample: its rank in the whole array is
  // ~ k * n / 64 Ki, below kCandCap up to n = 2^25 at k = 128 (beyond that the overflow fallback decides)
  const uint32_t per = 64;
  c->ensure_out(k);
  // The K winners, their count and the overflow flag are written by the last kernel straight into the
  // pinned (device-visible, coherent) host buffers: no memset, no D2H copies on the critical path.
  c->h_fcnt[1] = 0;  // overflow flag, only ever set by the kernel
  c->h_fcnt[2] = 0;
  launch_sample_threshold(d_keys, n, per, k, c->d_tau, c->d_fcnt, c->stream);
  launch_filter_keys(d_keys, n, c->d_tau, c->d_cand, c->d_fcnt, QueryCtx::kCandCap, c->stream);
  launch_batch_select_cand(c->d_cand, c->d_fcnt, QueryCtx::kCandCap, k, 1, c->h_out_rows, (uint32_t *)c->h_out_keys,
                           c->h_fcnt + 2, k, c->h_fcnt + 1, c->stream);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(c->stream));
  collect_profile(c);
  if (c->h_fcnt[1]) return false;  // candidate overflow: radix path
  const uint32_t got = std::min<uint32_t>(c->h_fcnt[2], k);
  if (got < std::min<uint32_t>(k, n)) return false;  // cannot happen (tau is an upper bound); be safe
  const uint32_t *k32 = reinterpret_cast<const uint32_t *>(c->h_out_keys);
  out.resize(got);
  for (uint32_t i = 0; i < got; i++) out[i] = Hit{c->h_out_rows[i], (uint64_t)k32[i]};
  std::sort(out.begin(), out.end(), [](const Hit &a, const Hit &b) { return a.key != b.key ? a.key < b.key : a.row < b.row; });
  return true;
}

// Small indexes (n <= 2^15): the whole selection is ONE workgroup's radix select over the keys (they sit in L2),
// winners written straight into pinned host memory -- one launch and one sync instead of four histogram levels,
// a collect pass and three copies (10 k rows x 128: 91 -> 59 us per query; at 100 k rows one CU is too slow).
static bool small_select(QueryCtx *c, const uint32_t *d_keys, uint32_t n, uint32_t k, std::vector<Hit> &out) {
  c->ensure_out(k);
  c->h_fcnt[2] = 0;
  launch_batch_select_keys(d_keys, n, n, k, 1, c->h_out_rows, (uint32_t *)c->h_out_keys, c->h_fcnt + 2, k, c->stream);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(c->stream));
  collect_profile(c);
  const uint32_t got = std::min<uint32_t>(c->h_fcnt[2], k);
  if (got < std::min<uint32_t>(k, n)) return false;
  const uint32_t *k32 = reinterpret_cast<const uint32_t *>(c->h_out_keys);
  out.resize(got);
  for (uint32_t i = 0; i < got; i++) out[i] = Hit{c->h_out_rows[i], (uint64_t)k32[i]};
  std::sort(out.begin(), out.end(), [](const Hit &a, const Hit &b) { return a.key != b.key ? a.key < b.key : a.row < b.row; });
  return true;
}

// The two one-sync paths above, split in halves for callers that keep several streams busy (the fused hybrid query):
// enqueue launches the selection on c->stream and returns which path it took (0: none applies -- use select_keys32);
// nothing is synchronised.  Once the caller has synchronised the stream, the winners sit UNSORTED in c->h_out_rows /
// c->h_out_keys (u32 keys) and collect says how many -- or false when the path has to be redone by select_keys32
// (candidate overflow).
int select_keys32_enqueue(QueryCtx *c, const uint32_t *d_keys, uint32_t n, uint32_t k) {
  if (!scan_tuning().filter_select || k == 0) return 0;
  if (k <= 1024 && n <= (1u << 15)) {
    c->ensure_out(k);
    c->h_fcnt[2] = 0;
    launch_batch_select_keys(d_keys, n, n, k, 1, c->h_out_rows, (uint32_t *)c->h_out_keys, c->h_fcnt + 2, k, c->stream);
    return 1;
  }
  if (k <= 32 && n >= (1u << 16)) {
    c->ensure_out(k);
    c->h_fcnt[1] = 0;
    c->h_fcnt[2] = 0;
    launch_sample_threshold(d_keys, n, 64, k, c->d_tau, c->d_fcnt, c->stream);
    launch_filter_keys(d_keys, n, c->d_tau, c->d_cand, c->d_fcnt, QueryCtx::kCandCap, c->stream);
    launch_batch_select_cand(c->d_cand, c->d_fcnt, QueryCtx::kCandCap, k, 1, c->h_out_rows, (uint32_t *)c->h_out_keys,
                             c->h_fcnt + 2, k, c->h_fcnt + 1, c->stream);
    return 2;
  }
  return 0;
}
bool select_keys32_collect(QueryCtx *c, int mode, uint32_t n, uint32_t k, uint32_t *got) {
  if (mode == 2 && c->h_fcnt[1]) return false;
  *got = std::min<uint32_t>(c->h_fcnt[2], k);
  return *got >= std::min<uint32_t>(k, n);
}

void select_keys32(QueryCtx *c, const uint32_t *d_keys, uint32_t n, uint32_t k, std::vector<Hit> &out) {
  if (k > 0 && k <= 1024 && n <= (1u << 15) && scan_tuning().filter_select && small_select(c, d_keys, n, k, out)) return;
  if (k > 0 && k <= 32 && n >= (1u << 16) && scan_tuning().filter_select && filter_select(c, d_keys, n, k, out)) return;
  radix_select(c, d_keys, 4, n, k, Bound(), out, nullptr);
}

void FlatIndex::select(QueryCtx *c, uint32_t n, uint32_t k, const Bound &lower, std::vector<Hit> &out, Bound *upper) {
  if (key_bytes == 4 && !lower.valid && !upper && k > 0 && k <= 1024 && n <= (1u << 15) && scan_tuning().filter_select) {
    if (small_select(c, c->d_keys, n, k, out)) return;
  }
  // measured on 10M keys (post-scan time, filter vs radix levels): k=10 57 vs 118 us, k=16 68 vs 124,
  // k=32 96 vs 121, k=64 131 vs 128, k=100 158 vs 123 -- the single-workgroup final select over
  // ~k*n/64Ki candidates is what grows
  if (key_bytes == 4 && !lower.valid && !upper && k > 0 && k <= 32 && n >= (1u << 16) && scan_tuning().filter_select) {
    if (filter_select(c, c->d_keys, n, k, out)) return;
  }
  radix_select(c, c->d_keys, key_bytes, n, k, lower, out, upper);
}

// Two-stage exact top-K over an fp16 shadow (FLOAT32 cosine indexes created with ScanTuning::shadow16).
//   1. scan the shadow with the fp16 query: d16(row), half the bytes of the fp32 scan;
//   2. tau = sampled upper bound of the K-th smallest d16 (sample_threshold_kernel);
//   3. keep every row with d16 <= tau + 2*eps, where eps bounds |d16 - d32| for unit-norm rows and query
//      (u = 2^-11, the unit roundoff of fp16):
//        |q16.x16 - q.x| <= |q16||x16 - x| + |q16 - q||x| <= u (1 + u) + u              = 9.8e-4
//        + 6.1e-5 * sqrt(dim)  if the dot unit flushes fp16 subnormals (|x_i| < 2^-14)  <= 1.95e-3 at dim 1024
//        + dim * 2^-24         fp32 accumulation of the products                        <= 6.1e-5
//      -- eps = 4e-3 covers it up to dim 1024.  The true top-K of d32 is inside that set: a row of the true
//      top-K has d32 <= K-th d32 <= K-th d16 + eps <= tau + eps, hence d16 <= tau + 2 eps;
//   4. re-score the survivors from the fp32 rows with the SAME gather kernel arithmetic as the full scan and
//      select the K best (distance, row): ids and distances are bit-identical to the one-stage path.
// Returns false (caller runs the full fp32 scan) when the survivors do not fit the candidate buffer.
bool FlatIndex::shadow8_query(const float *qf, int8_t *q8, float *sq_out, float *qn2_out, float *eps_out) const {
  float qmax = 0.0f;
  double qn2 = 0.0;
  for (size_t i = 0; i < dim; i++) {
    qmax = std::max(qmax, std::fabs(qf[i]));
    qn2 += (double)qf[i] * (double)qf[i];
  }
  if (!(qmax > 0.0f) || !(s_max_ > 0.0f) || s_bad_ || !std::isfinite(qn2) || !std::isfinite(n2_max_)) return false;
  const float sq = qmax / 127.0f;
  const float qn = (float)std::sqrt(qn2) * 1.000001f, xn = std::sqrt(n2_max_) * 1.0001f;
  for (size_t i = 0; i < dim; i++) q8[i] = (int8_t)std::min(127.0f, std::max(-127.0f, std::nearbyintf(qf[i] / sq)));
  const float rd = std::sqrt((float)dim);
  const float eps_dot = ((s_max_ * qn + sq * xn) * rd * 0.5f + 0.75f * s_max_ * sq * (float)dim) * 1.001f;
  const bool l2 = kmetric == KM_L2;
  // fp32 rounding on both sides (d-term accumulations, |x|^2 of the row, the final sums): d 2^-23 of the largest magnitude
  const float mag = l2 ? (qn + xn) * (qn + xn) : 1.0f + qn * xn;
  const float eps = (l2 ? 2.0f * eps_dot : eps_dot) + (float)dim * 1.2e-7f * mag + 1e-6f;
  if (!std::isfinite(eps)) return false;
  *sq_out = sq;
  *qn2_out = (float)qn2;
  *eps_out = eps;
  return true;
}

bool FlatIndex::two_stage_topk(QueryCtx *c, uint32_t n, uint32_t k, std::vector<Hit> &out) {
  TwoStageStats &st = two_stage_stats();
  st.v[TwoStageStats::ATTEMPTS]++;
  auto leave = [&st](int why) {
    st.v[why]++;
    return false;
  };
  if (dim > 1024) return leave(TwoStageStats::FB_SHAPE);
  float kSlack = 2.0f * 4e-3f;
  // shadow-typed copy of the normalised query behind the fp32 one
  const size_t q16_off = round_up(stride_ + 16, 16);  // upload_query sized the buffers for it
  const float *qf = reinterpret_cast<const float *>(c->h_query);
  const bool prof = scan_profile().enabled.load(std::memory_order_relaxed) != 0;
  c->ensure_keys(n);
  c->ensure_out(k);
  // the survivors' gather buffers at their largest right away: sized per query they were re-allocated (a device free, a
  // device allocation and two pinned-host allocations, tens of milliseconds) whenever a query kept more rows than any
  // query before it on this workspace
  c->ensure_gather(QueryCtx::kCandCap);
  if (shadow_ == 1) {
    uint16_t *q16 = reinterpret_cast<uint16_t *>(c->h_query + q16_off);
    memset(q16, 0, sstride_);
    for (size_t i = 0; i < dim; i++) q16[i] = f2h(qf[i]);
    HIP_CHECK(hipMemcpyAsync(c->d_query + q16_off, q16, sstride_, hipMemcpyHostToDevice, c->stream));
    if (prof) HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    launch_scan(d_shadow_, sstride_, (uint32_t)dim, KT_F16, KM_IP, 0, n, c->d_query + q16_off, c->d_keys, c->stream);
  } else {
    // int8 shadow: x = sx (x8 + ex), q = sq (q8 + eq), |ex_i|, |eq_i| <= 1/2, so
    //   |sx sq x8.q8 - x.q| <= sx (|q| + sq sqrt(d)/2) sqrt(d)/2 + sq (|x| + sx sqrt(d)/2) sqrt(d)/2 + sx sq d/4
    //                        = (sx |q| + sq |x|) sqrt(d)/2 + (3/4) sx sq d,   sx <= s_max, |x| <= n_max over all rows
    // (unit vectors: (sx + sq) sqrt(d)/2 + ...).  IP / cosine distance 1 - x.q: that band; L2 = |q|^2 + |x|^2 - 2 x.q:
    // twice the band, |x|^2 taken from the fp32 row at add time.
    float sq = 0.0f, qn2f = 0.0f, eps = 0.0f;
    int8_t *q8 = reinterpret_cast<int8_t *>(c->h_query + q16_off);
    memset(q8, 0, sstride_ + 16);
    if (!shadow8_query(qf, q8, &sq, &qn2f, &eps)) return leave(TwoStageStats::FB_QUERY_OR_BAND);
    memcpy(q8 + sstride_ + 4, &sq, 4);  // the extra chunk: {0, query scale, |q|^2, 0}
    memcpy(q8 + sstride_ + 8, &qn2f, 4);
    HIP_CHECK(hipMemcpyAsync(c->d_query + q16_off, q8, sstride_ + 16, hipMemcpyHostToDevice, c->stream));
    const bool l2 = kmetric == KM_L2;
    kSlack = 2.0f * eps;
    if (prof) HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    launch_scan(d_shadow_, sstride_, (uint32_t)dim, KT_I8, l2 ? KM_L2S : KM_IPS, 0, n, c->d_query + q16_off, c->d_keys, c->stream,
                d_sscale_);
  }
  if (prof) {
    HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    c->prof_rows = n;
    c->prof_bytes_per_row = shadow_ == 1 ? dim * 2 : dim + 8;
    c->prof_pending = true;
  }
  c->h_fcnt[1] = 0;
  c->h_fcnt[2] = 0;
  if ((shadow_ == 2 && k > 16) || k > 128) {
    // the int8 band is wide: with a sampled bound the survivors of K > 16 outgrow the candidate buffer (and so do those
    // of the fp16 band for K > 128), so tau is the EXACT K-th shadow distance here (radix levels over the shadow
    // keys, ~0.1 ms)
    std::vector<Hit> tmp;
    Bound kth;
    radix_select(c, c->d_keys, 4, n, k, Bound(), tmp, &kth);
    const float tau = key_to_dist((uint32_t)kth.key);
    HIP_CHECK(hipMemcpyAsync(c->d_tau, &tau, sizeof tau, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemsetAsync(c->d_fcnt, 0, 4 * sizeof(uint32_t), c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));  // (tau lives on this frame)
  } else if (shadow_ == 2) {
    // int8 band (2 eps ~ 0.76 sigma on the bench corpus): filtering with the SAMPLED bound + 2 eps would keep ~16 k rows
    // and the append path of the filter pass costs 0.1 ms.  Two cheap passes instead: keys <= sampled bound (~1.5 k),
    // their exact K-th = the K-th shadow distance of the whole index, then keys <= that + 2 eps (~350)
    launch_sample_threshold(c->d_keys, n, 64, k, c->d_tau, c->d_fcnt, c->stream);
    launch_filter_keys(c->d_keys, n, c->d_tau, c->d_cand, c->d_fcnt, QueryCtx::kCandCap, c->stream);
    // (a list of exactly K candidates keeps the sampled bound, which is then the K-th distance itself)
    launch_batch_threshold_cand(c->d_cand, c->d_fcnt, QueryCtx::kCandCap, k, 1, 1, c->d_tau, c->h_fcnt + 1, c->stream);
    HIP_CHECK(hipMemsetAsync(c->d_fcnt, 0, 4 * sizeof(uint32_t), c->stream));
  } else {
    launch_sample_threshold(c->d_keys, n, 64, k, c->d_tau, c->d_fcnt, c->stream);
  }
  launch_filter_keys(c->d_keys, n, c->d_tau, c->d_cand, c->d_fcnt, QueryCtx::kCandCap, c->stream, kSlack);
  HIP_CHECK(hipMemcpyAsync(c->h_fcnt + 3, c->d_fcnt, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  collect_profile(c);
  const uint32_t m = c->h_fcnt[3];
  if (c->h_fcnt[1]) return leave(TwoStageStats::FB_FIRST_PASS_OVERFLOW);  // the first int8 pass overflowed
  if (m > QueryCtx::kCandCap) return leave(TwoStageStats::FB_BAND_OVERFLOW);
  if (m < k) return leave(TwoStageStats::FB_TOO_FEW);
  launch_cand_rows(c->d_cand, c->d_fcnt, QueryCtx::kCandCap, c->d_ids, c->stream);
  launch_gather(d_rows_, stride_, (uint32_t)dim, ktype, kmetric, c->d_ids, m, c->d_query, c->d_dists, c->stream);
  launch_cand_set_keys(c->d_cand, c->d_dists, m, c->stream);
  launch_batch_select_cand(c->d_cand, c->d_fcnt, QueryCtx::kCandCap, k, 1, c->h_out_rows, (uint32_t *)c->h_out_keys,
                           c->h_fcnt + 2, k, c->h_fcnt + 1, c->stream);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(c->stream));
  if (c->h_fcnt[1] || c->h_fcnt[2] < k) return leave(TwoStageStats::FB_SELECT);
  st.v[TwoStageStats::OK]++;
  const uint32_t *k32 = reinterpret_cast<const uint32_t *>(c->h_out_keys);
  out.resize(k);
  for (uint32_t i = 0; i < k; i++) out[i] = Hit{c->h_out_rows[i], (uint64_t)k32[i]};
  std::sort(out.begin(), out.end(), [](const Hit &a, const Hit &b) { return a.key != b.key ? a.key < b.key : a.row < b.row; });
  return true;
}

static void sort_reply(VecSimQueryReply *r, VecSimQueryReply_Order order) {
  if (order == BY_ID)
    std::sort(r->results, r->results + r->len, [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return a.id < b.id; });
  else
    std::sort(r->results, r->results + r->len, [](const VecSimQueryResult &a, const VecSimQueryResult &b) {
      return score_id_before(a.score, a.id, b.score, b.id);
    });
}

// ---- the coalescer (flat_index.hpp) --------------------------------------------------------------------------------
// Which calls go through it: corpora whose scan is HBM-bound (below ~64 MiB a query is a handful of launch latencies and
// concurrent streams serve concurrent callers better), row shapes the multi-query kernel has, single-value indexes with
// 32-bit keys, and no two-stage shadow in use (that path has its own, four times cheaper, scan).
bool FlatIndex::mq_capable(size_t k) const {
  if (!k || k > 1024) return false;
  if ((shadow_ == 1 || shadow_ == 2) && scan_tuning().two_stage) return false;
  return scan_mq_supported(ktype, kmetric, (uint32_t)(stride_ / 16));
}
// ... unless it is the int8 shadow and the caller wants few neighbours: then the two-stage scan itself has a multi-query form
// (K <= 16: above that the single path takes the exact K-th shadow distance through radix levels, which has no batched form)
bool FlatIndex::shadow8_mq_capable(size_t k) const {
  if (shadow_ != 2 || !scan_tuning().two_stage || !scan_tuning().coalesce_shadow8 || multi || key_bytes != 4 || !k || k > 16) return false;
  if (s_bad_ || sstride_ != dim || !scan_mq_i8_supported((uint32_t)(sstride_ / 16))) return false;
  if (!batch_rescore_supported((uint32_t)(stride_ / 16))) return false;
  return __atomic_load_n(&n_rows_, __ATOMIC_RELAXED) >= (1u << 18);
}
bool FlatIndex::coalescible(size_t k) const {
  const ScanTuning &t = scan_tuning();
  if (!t.coalesce || (!mq_capable(k) && !shadow8_mq_capable(k))) return false;
  const uint32_t n = __atomic_load_n(&n_rows_, __ATOMIC_RELAXED);
  return (size_t)n * stride_ >= ((size_t)(t.coalesce_min_mib > 0 ? t.coalesce_min_mib : 0) << 20);
}

int FlatIndex::coalesce_linger_us() const {
  const int us = scan_tuning().coalesce_linger_us;
  if (us >= 0) return us;
  // (~6 TB/s over the bytes a pass reads: the int8 shadow's when it serves the queries)
  const size_t row_bytes = shadow_ == 2 && scan_tuning().two_stage ? sstride_ + 8 : stride_;
  const double pass_us = (double)__atomic_load_n(&n_rows_, __ATOMIC_RELAXED) * (double)row_bytes / 6.0e6;
  // (the wait ends as soon as the expected callers are back; the bound only matters when they are not: 8 % of a pass, at least
  // the ~50 us a caller needs to wake up, merge and fan out again -- 5 % / 20 us left two-shard handles alternating between a
  // group of five and a group of three)
  return (int)std::min(400.0, std::max(50.0, 0.08 * pass_us));
}

VecSimQueryReply *FlatIndex::topk(const void *query, size_t k, VecSimQueryParams *qp, VecSimQueryReply_Order order) {
  void *tctx = qp ? qp->timeoutCtx : nullptr;
  last_mode = STANDARD_KNN;
  if (!coalescible(k)) {
    flush_if_needed();
    std::shared_lock<std::shared_mutex> g(mu);
    return topk_locked(query, k, tctx, order);
  }
  if (tctx && timed_out(tctx)) return new_reply(0, VecSim_QueryReply_TimedOut);  // (expired on arrival: never queued)
  TopkJob job{query, k, tctx, order};
  job.owner_polls = true;
  std::vector<TopkJob *> batch;
  // what the caller gets when ITS timeout fired while the call sat in the queue or in somebody's pass
  auto finish = [&](VecSimQueryReply *r) -> VecSimQueryReply * {
    if (!tctx || !r || r->code != VecSim_QueryReply_OK || !timed_out(tctx)) return r;
    host_free(r->results);
    host_free(r);
    return new_reply(0, VecSim_QueryReply_TimedOut);
  };
  // how many calls a pass may hold: sixteen through the exact multi-query scan; kWidePass where the index's batches go
  // through a matrix-core filter pass + exact re-scoring (same bits, one corpus pass for all of them)
  const uint32_t cap = wide_pass_capable(k) ? kWidePass : kMqMaxQueries;
  {
    std::unique_lock<std::mutex> lk(co_.mu);
    co_.waiting.push_back(&job);
    if (co_.lingering) co_.cv_leader.notify_one();
    // sleep until a leader has answered this query, or it is this caller's turn to lead -- in slices of a millisecond when
    // the call carries a timeout context: a call that times out while it is QUEUED (a pass it is not part of takes 5-6 ms at
    // 10 M x 768) leaves the queue at once (reference src/util/timeout.h:70,89 polls every 100 iterations of its loops)
    auto ready = [&] { return job.done || (!co_.busy && !co_.waiting.empty() && co_.waiting.front() == &job); };
    if (!tctx) {
      co_.cv.wait(lk, ready);
    } else {
      while (!ready()) {  // (polled on arrival, then every millisecond)
        if (co_.cv.wait_for(lk, std::chrono::milliseconds(1), ready)) break;
        if (!job.taken) {   // (in a pass: its leader owns the job until the pass is done)
          lk.unlock();
          const bool expired = timed_out(tctx);  // (the host's callback: never under the coalescer's lock)
          lk.lock();
          if (expired && !job.taken && !job.done) {
            auto it = std::find(co_.waiting.begin(), co_.waiting.end(), &job);
            if (it != co_.waiting.end()) {
              const bool was_front = it == co_.waiting.begin();
              co_.waiting.erase(it);
              coalesce_stats().left_queue++;
              lk.unlock();
              if (was_front) co_.cv.notify_all();  // (whoever is first now may be due to lead)
              return new_reply(0, VecSim_QueryReply_TimedOut);
            }
          }
        }
      }
    }
    if (job.done) {
      lk.unlock();
      if (job.err) std::rethrow_exception(job.err);
      return finish(job.reply);
    }
    if (tctx) {  // this caller leads: its own timeout first (the passes do not poll a coalesced job's callback)
      lk.unlock();
      const bool expired = timed_out(tctx);
      lk.lock();
      if (expired) {
        co_.waiting.erase(std::find(co_.waiting.begin(), co_.waiting.end(), &job));
        lk.unlock();
        co_.cv.notify_all();
        return new_reply(0, VecSim_QueryReply_TimedOut);
      }
    }
    co_.busy = true;
    // the callers of the previous pass are on their way back with their next query: give them a moment, so that the
    // passes stay full instead of alternating between one early bird and everybody else
    const uint32_t expect = std::min<uint32_t>(co_.last_b, cap);
    if (co_.waiting.size() < expect) {
      const int us = coalesce_linger_us();
      if (us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        co_.lingering = true;
        co_.cv_leader.wait_for(lk, std::chrono::microseconds(us), [&] { return co_.waiting.size() >= expect; });
        co_.lingering = false;
        coalesce_stats().lingers++;
        coalesce_stats().linger_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      }
    }
    const size_t take = std::min<size_t>(co_.waiting.size(), cap);
    batch.assign(co_.waiting.begin(), co_.waiting.begin() + (long)take);
    for (TopkJob *j : batch) j->taken = true;
    co_.waiting.erase(co_.waiting.begin(), co_.waiting.begin() + (long)take);
  }
  std::exception_ptr err;
  try {
    topk_pass(batch.data(), batch.size());
  } catch (...) {
    err = std::current_exception();
  }
  {
    std::lock_guard<std::mutex> lk(co_.mu);
    for (TopkJob *j : batch) {
      if (err && !j->reply) j->err = err;
      j->done = true;
    }
    // how many callers there are: the ones this pass answered -- on their way back with their next query -- AND the ones
    // that arrived while it ran.  With the pass's own size alone two half-size groups can alternate for ever (each group's
    // leader sees its "expected" number right away and never waits for the other group to come back): half the throughput.
    co_.last_b = (uint32_t)std::min<size_t>(batch.size() + co_.waiting.size(), kWidePass);
    co_.busy = false;
  }
  co_.cv.notify_all();
  if (job.err) std::rethrow_exception(job.err);
  return finish(job.reply);
}

// set while a wide pass runs topk_batch on this thread: whatever that declines comes back through topk_pass in groups of
// sixteen and must take the exact scans there, not another wide pass
static thread_local bool tls_in_wide_pass = false;

size_t FlatIndex::wide_min() const {
  // (the early switch is for plain FLOAT32 indexes: next to a shadow the coalesced two-stage passes -- eight queries per 1.4 ms --
  // stay ahead until more callers queue than two of them hold)
  if (type != VecSimType_FLOAT32 || ((shadow_ != 0 || s8g_enabled()) && scan_tuning().two_stage)) return kMqMaxQueries + 1;
  // (... and for LARGE ones: the matrix-core pass carries ~0.3 ms of phases, thresholds and re-scoring that a 10 M x 768 corpus
  // amortises -- eight callers 1 655 QPS against 1 477 -- and a 2 M-row shard does not: eight such shards on one device fell
  // from 888 to 390 QPS with the early switch, profiles/r04_bench_8shards_one_device*.json)
  if ((uint64_t)__atomic_load_n(&n_rows_, __ATOMIC_RELAXED) * stride_ < (16ull << 30)) return kMqMaxQueries + 1;
  const int v = scan_tuning().coalesce_wide_min;
  return v < 2 ? 2 : (size_t)v;
}

// More than sixteen calls in one pass (or nine and more, knob coalesce_wide_min): their queries side by side through topk_batch -- a matrix-core filter pass over the
// corpus for all of them, exact re-scoring, the usual exact selection; every caller takes the leading K of its winners.
// Jobs the batch cannot hold (K beyond its limit) and whatever it declines are answered sixteen per exact pass.
void FlatIndex::topk_pass_wide(TopkJob *const *jobs, size_t n_jobs) {
  std::vector<TopkJob *> wide, rest;
  for (size_t i = 0; i < n_jobs; i++) {
    TopkJob *j = jobs[i];
    if (!j->owner_polls && timed_out(j->tctx)) j->reply = new_reply(0, VecSim_QueryReply_TimedOut);
    else if (!j->k) j->reply = new_reply(0, VecSim_QueryReply_OK);
    else (wide_pass_capable(j->k) ? wide : rest).push_back(j);
  }
  if (wide.size() < wide_min()) {
    rest.insert(rest.end(), wide.begin(), wide.end());
    wide.clear();
  }
  for (size_t at = 0; at < rest.size(); at += kMqMaxQueries) topk_pass(rest.data() + at, std::min<size_t>(kMqMaxQueries, rest.size() - at));
  if (wide.empty()) return;
  coalesce_stats().passes++;
  coalesce_stats().queries += wide.size();
  coalesce_stats().wide_passes++;
  coalesce_stats().wide_queries += wide.size();
  size_t kmax = 0;
  for (TopkJob *j : wide) kmax = std::max(kmax, j->k);
  std::vector<uint8_t> qbuf(wide.size() * elem_bytes_);
  std::vector<size_t> k_each(wide.size()), ids(wide.size() * kmax), cnt(wide.size());
  std::vector<double> sc(wide.size() * kmax);
  for (size_t i = 0; i < wide.size(); i++) {
    memcpy(qbuf.data() + i * elem_bytes_, wide[i]->query, elem_bytes_);
    k_each[i] = wide[i]->k;
  }
  {
    struct Flag {
      Flag() { tls_in_wide_pass = true; }
      ~Flag() { tls_in_wide_pass = false; }
    } in_wide;
    topk_batch(qbuf.data(), wide.size(), kmax, ids.data(), sc.data(), cnt.data(), k_each.data());
  }
  for (size_t i = 0; i < wide.size(); i++) {
    TopkJob *j = wide[i];
    if (!j->owner_polls && timed_out(j->tctx)) {
      j->reply = new_reply(0, VecSim_QueryReply_TimedOut);
      continue;
    }
    VecSimQueryReply *r = new_reply(cnt[i], VecSim_QueryReply_OK);
    for (size_t t = 0; t < cnt[i]; t++) r->results[t] = VecSimQueryResult{ids[i * kmax + t], sc[i * kmax + t]};
    sort_reply(r, j->order);
    j->reply = r;
  }
}

void FlatIndex::topk_pass(TopkJob *const *jobs, size_t n_jobs) {
  if (!n_jobs) return;
  // nine and more callers: the matrix-core pass (4.9 ms per 10 M x 768 corpus whatever the number of queries) beats the exact
  // multi-query scan's sixteen-query form (5.8 ms, VALU-bound) -- where the index's batches are exact; see wide_min()
  bool wide = n_jobs > kMqMaxQueries;
  if (!wide && n_jobs >= wide_min() && !tls_in_wide_pass) {
    wide = true;
    for (size_t i = 0; i < n_jobs && wide; i++) wide = wide_pass_capable(jobs[i]->k);
  }
  if (wide) {
    topk_pass_wide(jobs, n_jobs);
    return;
  }
  coalesce_stats().passes++;
  coalesce_stats().queries += n_jobs;
  flush_if_needed();
  std::shared_lock<std::shared_mutex> g(mu);
  if (n_jobs == 1) {  // nobody to share the pass with: the single-query path, as if there were no coalescer
    jobs[0]->reply = topk_locked(jobs[0]->query, jobs[0]->k, jobs[0]->tctx, jobs[0]->order);
    return;
  }
  const uint32_t n = n_rows_;
  // jobs the multi-query pass can serve together; everything else (timed out, nothing to return) is answered right here
  TopkJob *live[kMqMaxQueries];
  size_t n_live = 0;
  for (size_t i = 0; i < n_jobs; i++) {
    TopkJob *j = jobs[i];
    if (!j->owner_polls && timed_out(j->tctx)) j->reply = new_reply(0, VecSim_QueryReply_TimedOut);
    else if (!n || !j->k) j->reply = new_reply(0, VecSim_QueryReply_OK);
    else live[n_live++] = j;
  }
  bool mq = n_live >= 2, mq8 = n_live >= 2;
  for (size_t i = 0; i < n_live; i++) {
    mq = mq && mq_capable(live[i]->k);
    mq8 = mq8 && shadow8_mq_capable(live[i]->k);
  }
  if (mq) {
    topk_pass_mq(live, n_live, n);
    return;
  }
  if (mq8) {  // (the shadow kernel holds eight queries)
    for (size_t at = 0; at < n_live; at += 8) {
      const size_t cnt = std::min<size_t>(8, n_live - at);
      if (cnt == 1) live[at]->reply = topk_locked(live[at]->query, live[at]->k, live[at]->tctx, live[at]->order);
      else topk_pass_mq_shadow8(live + at, cnt, n);
    }
    return;
  }
  for (size_t i = 0; i < n_live; i++) live[i]->reply = topk_locked(live[i]->query, live[i]->k, live[i]->tctx, live[i]->order);
}

// >= 2 queries, one pass: the multi-query scan writes one key array per query (bit for bit the single-query scan's), then
// the same selection as a single query's -- for every query at once where the path has a batched form (small K: sampled
// bound, one filter pass, one select workgroup per query; short arrays: one select workgroup per query), query by query
// through the radix levels otherwise.  K differs per caller: the selection runs with the largest, every caller gets the
// leading K of its sorted winners (the composite (key, row) order makes the top-K a prefix of the top-K').
void FlatIndex::topk_pass_mq(TopkJob *const *jobs, size_t nj, uint32_t n) {
  HIP_CHECK(hipSetDevice(device));
  CtxLease c(device);
  // INT8 / UINT8: every padded query is followed by its extra chunk {sum q^2, |q|} (as upload_query lays it out)
  const bool int_type = type == VecSimType_INT8 || type == VecSimType_UINT8;
  const size_t qstride = stride_ + (int_type ? 16 : 0);
  const size_t kw = key_bytes / 4;  // u32 words per key
  c->ensure_mq(nj * qstride);
  memset(c->h_mq_queries, 0, nj * qstride);
  uint32_t kmax = 0;
  for (size_t b = 0; b < nj; b++) {
    uint8_t *dst = c->h_mq_queries + b * qstride;
    memcpy(dst, jobs[b]->query, elem_bytes_);
    if (metric == VecSimMetric_Cosine) normalize_host(dst);
    if (int_type) {
      long long qq = 0;
      for (size_t i = 0; i < dim; i++) {
        const int a = type == VecSimType_INT8 ? (int)((const int8_t *)jobs[b]->query)[i] : (int)((const uint8_t *)jobs[b]->query)[i];
        qq += (long long)a * a;
      }
      uint32_t extra[4] = {(uint32_t)qq, 0, 0, 0};
      const float qn = sqrtf((float)qq);
      memcpy(&extra[1], &qn, 4);
      memcpy(dst + stride_, extra, 16);
    }
    kmax = std::max<uint32_t>(kmax, (uint32_t)std::min<size_t>(jobs[b]->k, n));
  }
  HIP_CHECK(hipMemcpyAsync(c->d_mq_queries, c->h_mq_queries, nj * qstride, hipMemcpyHostToDevice, c->stream));
  const uint32_t ld = (uint32_t)round_up(n, 1024);
  c->ensure_keys(nj * (size_t)ld * kw);
  c->ensure_out(nj * (size_t)kmax);
  const bool prof = scan_profile().enabled.load(std::memory_order_relaxed) != 0;
  HIP_CHECK(hipEventRecord(c->ev0, c->stream));
  if (!launch_scan_mq(d_rows_, stride_, ktype, kmetric, 0, n, c->d_mq_queries, qstride, (uint32_t)nj, c->d_keys, ld, c->stream))
    throw std::runtime_error("multi-query scan refused a row shape the coalescer was gated on");
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventRecord(c->ev1, c->stream));
  // ---- selection
  std::vector<Hit> hits[kMqMaxQueries];
  bool have[kMqMaxQueries] = {false};
  int mode = 0;
  uint32_t *out_keys32 = reinterpret_cast<uint32_t *>(c->h_out_keys);
  const bool batched_select = scan_tuning().filter_select && key_bytes == 4 && !multi;  // (the batched forms take 32-bit keys)
  if (batched_select && kmax <= 1024 && n <= (1u << 15)) {
    mode = 1;
    for (size_t b = 0; b < nj; b++) c->h_mq_n[b] = 0;
    launch_batch_select_keys(c->d_keys, ld, n, kmax, (uint32_t)nj, c->h_out_rows, out_keys32, c->h_mq_n, kmax, c->stream);
  } else if (batched_select && kmax <= 32 && n >= (1u << 16)) {
    mode = 2;
    for (size_t b = 0; b < nj; b++) c->h_mq_n[b] = c->h_mq_over[b] = 0;
    launch_sample_threshold_batch(c->d_keys, ld, n, 64, kmax, (uint32_t)nj, c->d_mq_tau, c->d_mq_cnt, c->stream);
    launch_filter_keys_batch(c->d_keys, ld, n, (uint32_t)nj, c->d_mq_tau, c->d_mq_cand, c->d_mq_cnt, QueryCtx::kCandCap, c->stream);
    launch_batch_select_cand(c->d_mq_cand, c->d_mq_cnt, QueryCtx::kCandCap, kmax, (uint32_t)nj, c->h_out_rows, out_keys32,
                             c->h_mq_n, kmax, c->h_mq_over, c->stream);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(c->stream));
  {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) {
      coalesce_stats().mq_device_ns += (uint64_t)((double)ms * 1e6);
      if (prof) {  // one pass over the corpus, whatever the number of queries it served
        ScanProfile &pf = scan_profile();
        pf.launches++;
        pf.bytes += (uint64_t)n * elem_bytes_;
        pf.nanos += (uint64_t)((double)ms * 1e6);
      }
    }
  }
  coalesce_stats().mq_passes++;
  coalesce_stats().mq_queries += nj;
  if (mode) {
    for (size_t b = 0; b < nj; b++) {
      const uint32_t got = std::min<uint32_t>(c->h_mq_n[b], kmax);
      if ((mode == 2 && c->h_mq_over[b]) || got < std::min<uint32_t>(kmax, n)) continue;  // redone below
      hits[b].resize(got);
      for (uint32_t i = 0; i < got; i++) hits[b][i] = Hit{c->h_out_rows[b * kmax + i], (uint64_t)out_keys32[b * kmax + i]};
      std::sort(hits[b].begin(), hits[b].end(), [](const Hit &x, const Hit &y) { return x.key != y.key ? x.key < y.key : x.row < y.row; });
      have[b] = true;
    }
  }
  if (multi) {
    // multi-value: per query the walk of topk_locked over ITS key array -- batches in ascending composite order, the first
    // occurrence of a label is its best vector
    for (size_t b = 0; b < nj; b++) {
      TopkJob *j = jobs[b];
      std::vector<VecSimQueryResult> res;
      const bool ok = multi_walk(c.c, c->d_keys + b * (size_t)ld * kw, n, j->k, j->tctx, res);
      if (!ok) {
        j->reply = new_reply(0, VecSim_QueryReply_TimedOut);
        continue;
      }
      VecSimQueryReply *r = new_reply(res.size(), VecSim_QueryReply_OK);
      if (!res.empty()) memcpy(r->results, res.data(), res.size() * sizeof(VecSimQueryResult));
      sort_reply(r, j->order);
      j->reply = r;
    }
    return;
  }
  for (size_t b = 0; b < nj; b++) {
    if (have[b]) continue;
    if (mode) coalesce_stats().mq_redo++;
    radix_select(c.c, c->d_keys + b * (size_t)ld * kw, key_bytes, n, (uint32_t)std::min<size_t>(jobs[b]->k, n), Bound(), hits[b], nullptr);
  }
  for (size_t b = 0; b < nj; b++) {
    TopkJob *j = jobs[b];
    if (!j->owner_polls && timed_out(j->tctx)) {
      j->reply = new_reply(0, VecSim_QueryReply_TimedOut);
      continue;
    }
    const size_t take = std::min<size_t>(hits[b].size(), j->k);
    VecSimQueryReply *r = new_reply(take, VecSim_QueryReply_OK);
    for (size_t i = 0; i < take; i++) r->results[i] = VecSimQueryResult{(size_t)label_at(hits[b][i].row), score_of(hits[b][i].key)};
    sort_reply(r, j->order);
    j->reply = r;
  }
}

// multi-value top-K over one key array: batches in ascending composite (key, row) order, the first occurrence of a label is
// its best vector.  false = the caller's timeout fired.
bool FlatIndex::multi_walk(QueryCtx *c, const uint32_t *d_keys, uint32_t n, size_t k, void *tctx, std::vector<VecSimQueryResult> &res) {
  std::unordered_map<uint64_t, char> seen;
  std::vector<Hit> hits;
  Bound lower;
  uint32_t consumed = 0;
  const size_t want = std::min<size_t>(k, n);
  while (res.size() < want && consumed < n) {
    const uint32_t ask = (uint32_t)std::min<size_t>(n - consumed, std::max<size_t>((want - res.size()) * 2, 16));
    Bound bound;
    radix_select(c, d_keys, key_bytes, n, ask, lower, hits, &bound);
    if (hits.empty()) break;
    consumed += (uint32_t)hits.size();
    for (const Hit &h : hits) {
      const uint64_t lab = label_at(h.row);
      if (res.size() < want && seen.emplace(lab, 1).second) res.push_back(VecSimQueryResult{(size_t)lab, score_of(h.key)});
    }
    lower = bound;
    if (timed_out(tctx)) return false;
  }
  return true;
}

// The two-stage exact scan (two_stage_topk, int8 shadow) for 2 .. 8 queries at once: one multi-query pass over the shadow
// writes every query's shadow keys; per query: sampled bound -> candidates -> their exact K-th shadow distance + the query's
// own 2 eps -> candidates inside the band -> EXACT keys from the fp32 rows (batch_rescore_kernel: the scan's arithmetic) ->
// select.  Every step is the batched form of the single path's; a query whose lists overflow, or that cannot be quantised,
// is answered on the single path.  K <= 16 (the selection runs with the largest K of the pass: a wider, still valid, band).
void FlatIndex::topk_pass_mq_shadow8(TopkJob *const *jobs, size_t nj, uint32_t n) {
  HIP_CHECK(hipSetDevice(device));
  CtxLease c(device);
  const size_t off_q8 = nj * stride_, off_qx = off_q8 + nj * sstride_, off_slack = off_qx + nj * 2 * sizeof(float);
  const size_t total = round_up(off_slack + nj * sizeof(float), 16);
  c->ensure_mq(total);
  memset(c->h_mq_queries, 0, total);
  bool ok[8] = {false};
  uint32_t kmax = 0;
  for (size_t b = 0; b < nj; b++) {
    uint8_t *dst = c->h_mq_queries + b * stride_;
    memcpy(dst, jobs[b]->query, elem_bytes_);
    if (metric == VecSimMetric_Cosine) normalize_host(dst);
    float sq = 1.0f, qn2 = 0.0f, eps = 0.0f;
    ok[b] = shadow8_query(reinterpret_cast<const float *>(dst), reinterpret_cast<int8_t *>(c->h_mq_queries + off_q8 + b * sstride_),
                          &sq, &qn2, &eps);
    float *qx = reinterpret_cast<float *>(c->h_mq_queries + off_qx) + 2 * b;
    qx[0] = sq;
    qx[1] = qn2;
    reinterpret_cast<float *>(c->h_mq_queries + off_slack)[b] = ok[b] ? 2.0f * eps : 0.0f;
    kmax = std::max<uint32_t>(kmax, (uint32_t)std::min<size_t>(jobs[b]->k, n));
  }
  HIP_CHECK(hipMemcpyAsync(c->d_mq_queries, c->h_mq_queries, total, hipMemcpyHostToDevice, c->stream));
  const uint32_t ld = (uint32_t)round_up(n, 1024);
  c->ensure_keys(nj * (size_t)ld);
  c->ensure_out(nj * (size_t)kmax);
  const bool prof = scan_profile().enabled.load(std::memory_order_relaxed) != 0;
  const float *d_slack = reinterpret_cast<const float *>(c->d_mq_queries + off_slack);
  HIP_CHECK(hipEventRecord(c->ev0, c->stream));
  if (!launch_scan_mq_i8(d_shadow_, sstride_, kmetric == KM_L2 ? KM_L2S : KM_IPS, 0, n, d_sscale_, c->d_mq_queries + off_q8, sstride_,
                         reinterpret_cast<const float *>(c->d_mq_queries + off_qx), (uint32_t)nj, c->d_keys, ld, c->stream))
    throw std::runtime_error("multi-query shadow scan refused a row shape the coalescer was gated on");
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventRecord(c->ev1, c->stream));
  uint32_t *out_keys32 = reinterpret_cast<uint32_t *>(c->h_out_keys);
  for (size_t b = 0; b < nj; b++) c->h_mq_n[b] = c->h_mq_over[b] = 0;
  // keys <= sampled bound -> their exact K-th + 2 eps -> keys inside the band
  launch_sample_threshold_batch(c->d_keys, ld, n, 64, kmax, (uint32_t)nj, c->d_mq_tau, c->d_mq_cnt, c->stream);
  launch_filter_keys_batch(c->d_keys, ld, n, (uint32_t)nj, c->d_mq_tau, c->d_mq_cand, c->d_mq_cnt, QueryCtx::kCandCap, c->stream);
  // (a list of exactly K candidates keeps the sampled bound, which is then the K-th distance itself: the band is added by
  // the filter, whichever bound it gets)
  launch_batch_threshold_cand(c->d_mq_cand, c->d_mq_cnt, QueryCtx::kCandCap, kmax, (uint32_t)nj, (uint32_t)nj, c->d_mq_tau,
                              c->h_mq_over, c->stream);
  HIP_CHECK(hipMemsetAsync(c->d_mq_cnt, 0, nj * sizeof(uint32_t), c->stream));
  launch_filter_keys_batch(c->d_keys, ld, n, (uint32_t)nj, c->d_mq_tau, c->d_mq_cand, c->d_mq_cnt, QueryCtx::kCandCap, c->stream,
                           d_slack);
  // exact keys of the survivors (lists that overflowed are skipped: the select flags them)
  if (!launch_batch_rescore(d_rows_, stride_, n, c->d_mq_queries, stride_, c->d_mq_cand, c->d_mq_cnt, QueryCtx::kCandCap, (uint32_t)nj,
                            nullptr, c->stream, KT_F32, kmetric == KM_L2 ? KM_L2 : KM_IP))
    throw std::runtime_error("coalesced two-stage pass: the re-scoring kernel refused a row shape the route was gated on");
  launch_batch_select_cand(c->d_mq_cand, c->d_mq_cnt, QueryCtx::kCandCap, kmax, (uint32_t)nj, c->h_out_rows, out_keys32, c->h_mq_n,
                           kmax, c->h_mq_over, c->stream);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(c->stream));
  {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) {
      coalesce_stats().mq_device_ns += (uint64_t)((double)ms * 1e6);
      if (prof) {
        ScanProfile &pf = scan_profile();
        pf.launches++;
        pf.bytes += (uint64_t)n * (dim + 8);
        pf.nanos += (uint64_t)((double)ms * 1e6);
      }
    }
  }
  coalesce_stats().mq_passes++;
  coalesce_stats().mq_queries += nj;
  TwoStageStats &st = two_stage_stats();
  for (size_t b = 0; b < nj; b++) {
    TopkJob *j = jobs[b];
    if (timed_out(j->tctx)) {
      j->reply = new_reply(0, VecSim_QueryReply_TimedOut);
      continue;
    }
    const uint32_t got = std::min<uint32_t>(c->h_mq_n[b], kmax), want = (uint32_t)std::min<size_t>(j->k, n);
    st.v[TwoStageStats::ATTEMPTS]++;
    if (!ok[b] || c->h_mq_over[b] || got < std::min<uint32_t>(kmax, n)) {  // answered on the single path (which counts its own way out)
      st.v[TwoStageStats::ATTEMPTS]--;
      coalesce_stats().mq_redo++;
      j->reply = topk_locked(j->query, j->k, j->tctx, j->order);
      continue;
    }
    st.v[TwoStageStats::OK]++;
    std::vector<Hit> hits(got);
    for (uint32_t i = 0; i < got; i++) hits[i] = Hit{c->h_out_rows[b * kmax + i], (uint64_t)out_keys32[b * kmax + i]};
    std::sort(hits.begin(), hits.end(), [](const Hit &x, const Hit &y) { return x.key != y.key ? x.key < y.key : x.row < y.row; });
    const size_t take = std::min<size_t>(hits.size(), want);
    VecSimQueryReply *r = new_reply(take, VecSim_QueryReply_OK);
    for (size_t i = 0; i < take; i++) r->results[i] = VecSimQueryResult{(size_t)label_at(hits[i].row), score_of(hits[i].key)};
    sort_reply(r, j->order);
    j->reply = r;
  }
}

VecSimQueryReply *FlatIndex::topk_locked(const void *query, size_t k, void *tctx, VecSimQueryReply_Order order) {
  // the callback is polled at least once, even for an index smaller than one block (App. B-7)
  if (timed_out(tctx)) return new_reply(0, VecSim_QueryReply_TimedOut);
  const uint32_t n = n_rows_;
  if (!n || !k) return new_reply(0, VecSim_QueryReply_OK);
  HIP_CHECK(hipSetDevice(device));
  CtxLease c(device);
  upload_query(c.c, query, true);
  std::vector<Hit> hits;
  // fp16 shadow: error-bounded filter + exact fp32 re-scoring of the survivors; falls back to the full scan
  const bool two_stage = (shadow_ == 1 || shadow_ == 2) && scan_tuning().two_stage && k <= 1024 && n >= (1u << 18) && two_stage_topk(c.c, n, (uint32_t)std::min<size_t>(k, n), hits);
  if (!two_stage) scan_all(c.c, n);
  std::vector<VecSimQueryResult> res;
  if (!multi) {
    uint32_t kk = (uint32_t)std::min<size_t>(k, n);
    if (!two_stage) select(c.c, n, kk, Bound(), hits, nullptr);
    if (timed_out(tctx)) return new_reply(0, VecSim_QueryReply_TimedOut);
    res.reserve(hits.size());
    for (const Hit &h : hits) res.push_back(VecSimQueryResult{(size_t)label_at(h.row), score_of(h.key)});
  } else {
    // multi-value: walk batches in ascending composite order, first occurrence of a label is its best
    if (!multi_walk(c.c, c->d_keys, n, k, tctx, res)) return new_reply(0, VecSim_QueryReply_TimedOut);
  }
  VecSimQueryReply *r = new_reply(res.size(), VecSim_QueryReply_OK);
  if (!res.empty()) memcpy(r->results, res.data(), res.size() * sizeof(VecSimQueryResult));
  sort_reply(r, order);
  return r;
}

VecSimQueryReply *FlatIndex::range(const void *query, double radius, VecSimQueryParams *qp, VecSimQueryReply_Order order) {
  void *tctx = qp ? qp->timeoutCtx : nullptr;
  last_mode = RANGE_QUERY;
  flush_if_needed();
  std::shared_lock<std::shared_mutex> g(mu);
  if (timed_out(tctx)) return new_reply(0, VecSim_QueryReply_TimedOut);
  const uint32_t n = n_rows_;
  if (!n || std::isnan(radius)) return new_reply(0, VecSim_QueryReply_OK);
  HIP_CHECK(hipSetDevice(device));
  CtxLease c(device);
  upload_query(c.c, query, true);
  scan_all(c.c, n);
  // largest fp32 value whose widening is <= radius: `(double)dist <= radius`, inclusive (App. B-12)
  uint64_t max_key;
  if (key_bytes == 8) {
    max_key = dist64_to_key(radius);
  } else {
    float fr = (float)radius;
    if ((double)fr > radius) fr = std::nextafterf(fr, -INFINITY);
    max_key = dist_to_key(fr);
  }
  HIP_CHECK(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(uint32_t), c->stream));
  launch_range(c->d_keys, key_bytes, n, max_key, 0, c->d_counters, nullptr, nullptr, 0, c->stream);
  HIP_CHECK(hipMemcpyAsync(c->h_counters, c->d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  collect_profile(c.c);
  uint32_t cnt = c->h_counters[0];
  if (timed_out(tctx)) return new_reply(0, VecSim_QueryReply_TimedOut);
  if (!cnt) return new_reply(0, VecSim_QueryReply_OK);
  c->ensure_out(cnt);
  HIP_CHECK(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(uint32_t), c->stream));
  launch_range(c->d_keys, key_bytes, n, max_key, 1, c->d_counters, c->d_out_rows, c->d_out_keys, (uint32_t)c->out_cap, c->stream);
  HIP_CHECK(hipMemcpyAsync(c->h_out_rows, c->d_out_rows, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipMemcpyAsync(c->h_out_keys, c->d_out_keys, (size_t)cnt * key_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  std::vector<VecSimQueryResult> res;
  res.reserve(cnt);
  const uint32_t *rk32 = reinterpret_cast<const uint32_t *>(c->h_out_keys);
  const uint64_t *rk64 = c->h_out_keys;
  auto range_score = [&](uint32_t i) { return key_bytes == 8 ? key_to_dist64(rk64[i]) : (double)key_to_dist(rk32[i]); };
  if (!multi) {
    for (uint32_t i = 0; i < cnt; i++)
      res.push_back(VecSimQueryResult{(size_t)label_at(c->h_out_rows[i]), range_score(i)});
  } else {
    std::unordered_map<uint64_t, size_t> best;
    for (uint32_t i = 0; i < cnt; i++) {
      uint64_t lab = label_at(c->h_out_rows[i]);
      double d = range_score(i);
      auto it = best.find(lab);
      if (it == best.end()) {
        best[lab] = res.size();
        res.push_back(VecSimQueryResult{(size_t)lab, d});
      } else if (d < res[it->second].score) res[it->second].score = d;
    }
  }
  VecSimQueryReply *r = new_reply(res.size(), VecSim_QueryReply_OK);
  memcpy(r->results, res.data(), res.size() * sizeof(VecSimQueryResult));
  sort_reply(r, order);
  return r;
}

void FlatIndex::gather(QueryCtx *c, const size_t *labels, size_t m, double *out) {
  // the caller holds (at least) the shared lock and has uploaded the query into c->d_query
  std::vector<uint32_t> rows, ids;
  std::vector<uint32_t> first(m), count(m);
  ids.reserve(m);
  for (size_t i = 0; i < m; i++) {
    rows_of(labels[i], rows);
    first[i] = (uint32_t)ids.size();
    uint32_t cnt = 0;
    for (uint32_t r : rows)
      if (r < n_rows_) { ids.push_back(r); cnt++; }
    if (!cnt) { ids.push_back(0xFFFFFFFFu); cnt = 1; }
    count[i] = cnt;
  }
  size_t t = ids.size();
  c->ensure_gather(t);
  memcpy(c->h_ids, ids.data(), t * sizeof(uint32_t));
  HIP_CHECK(hipMemcpyAsync(c->d_ids, c->h_ids, t * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  launch_gather(d_rows_, stride_, (uint32_t)dim, ktype, kmetric, c->d_ids, (uint32_t)t, c->d_query, c->d_dists, c->stream);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(c->h_dists, c->d_dists, t * (size_t)key_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  const double *h_d64 = reinterpret_cast<const double *>(c->h_dists);
  for (size_t i = 0; i < m; i++) {
    double best = NAN;
    for (uint32_t j = 0; j < count[i]; j++) {
      double d = key_bytes == 8 ? h_d64[first[i] + j] : (double)c->h_dists[first[i] + j];
      if (std::isnan(best) || d < best) best = d;  // multi-value: minimum over the label's vectors
    }
    out[i] = best;
  }
}

// Per-thread workspace for GetDistanceFrom_Unsafe: the caller loops over candidates with the same
// (pre-normalised) blob, so the query stays resident on the device between calls.
struct TlsCtx {
  QueryCtx *c = nullptr;
  ~TlsCtx() {
    if (c) CtxPool::get().release(c);
  }
};
static thread_local TlsCtx tls_adhoc;

double FlatIndex::distance_from(size_t label, const void *blob) {
  flush_if_needed();
  std::shared_lock<std::shared_mutex> g(mu);
  HIP_CHECK(hipSetDevice(device));
  if (tls_adhoc.c && tls_adhoc.c->device != device) {
    CtxPool::get().release(tls_adhoc.c);
    tls_adhoc.c = nullptr;
  }
  if (!tls_adhoc.c) tls_adhoc.c = CtxPool::get().acquire(device);
  QueryCtx *c = tls_adhoc.c;
  c->ensure_query(stride_ + 16);
  if (!(c->cached_query_owner == uid && c->cached_query_len == elem_bytes_ && memcmp(c->h_query, blob, elem_bytes_) == 0)) {
    upload_query(c, blob, false);  // blob is already normalised by the caller (hybrid_reader.c:295-305)
    c->cached_query_owner = uid;
    c->cached_query_len = elem_bytes_;
  }
  double out;
  gather(c, &label, 1, &out);
  return out;
}

// VecSimIndex_PreferAdHocSearch for a brute-force index ([upstream-memory D6]; the four decision
// points the reference pins are listed in SURVEY.md 8 a6).
bool FlatIndex::prefer_adhoc_rule(size_t N, size_t labels, size_t d, size_t subset) {
  if (subset > N) subset = N;
  // the ratio is over LABELS (a multi-value index holds more vectors than documents), kept in float and compared
  // with double literals -- so exactly-on-threshold ratios fall where the float lands
  const float r = N ? (float)subset / (float)labels : 0.0f;
  if (N <= 5500) return true;
  if (d <= 300) {
    if (r <= 0.15) return true;
    if (r <= 0.35) return d <= 75 ? false : N <= 550000;
    return false;
  }
  if (r <= 0.55) return true;
  if (d <= 750) return false;
  return r <= 0.75;
}

bool FlatIndex::prefer_adhoc(size_t subset, size_t k, bool initial_check) {
  (void)k;
  const bool res = prefer_adhoc_rule(size(), label_count(), dim, subset);
  last_mode = res ? (initial_check ? HYBRID_ADHOC_BF : HYBRID_BATCHES_TO_ADHOC_BF) : HYBRID_BATCHES;
  return res;
}

}  // namespace rsgpu
