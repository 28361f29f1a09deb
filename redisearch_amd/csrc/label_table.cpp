// label_table.cpp -- see label_table.hpp.  Host bookkeeping of the writers + the few device operations that keep the kernels'
// copy current (corpus_kernels.hip: fill / scatter / decode).
#include "label_table.hpp"

#include <algorithm>

namespace rsgpu {

namespace {
constexpr uint64_t kSmallBase = 1ull << 20;  // labels that start below this are addressed from 0
constexpr size_t kMinLimit = 1ull << 20;     // entries (4 MiB) every index may spend on its direct table
constexpr size_t kMaxLimit = 1ull << 31;
}  // namespace

LabelTable::LabelTable(bool multi, size_t row_bytes, const LabelVec *row_label, size_t *host_bytes)
    : multi_(multi),
      row_bytes_(row_bytes),
      row_label_(row_label),
      host_bytes_(host_bytes) {}

LabelTable::~LabelTable() {
  free_direct();
  free_hash();
  if (h_pin_) (void)hipHostFree(h_pin_);
  if (d_pend_) (void)hipFree(d_pend_);
}

void *LabelTable::hook_calloc(size_t n, size_t sz) {
  void *p = hooks().mem.callocFunction(n, sz);
  if (!p) throw std::bad_alloc();
  if (host_bytes_) *host_bytes_ += n * sz;
  return p;
}
void LabelTable::hook_free(void *p, size_t bytes) {
  if (!p) return;
  hooks().mem.freeFunction(p);
  if (host_bytes_) *host_bytes_ -= bytes;
}

void LabelTable::free_direct() {
  hook_free(enc_, enc_cap_ * sizeof(uint32_t));
  hook_free(nxt_, nxt_cap_ * sizeof(uint32_t));
  enc_ = nxt_ = nullptr;
  enc_cap_ = nxt_cap_ = 0;
  if (d_row_of_) (void)hipFree(d_row_of_);
  if (d_next_) (void)hipFree(d_next_);
  d_row_of_ = d_next_ = nullptr;
  d_cap_ = d_next_cap_ = 0;
  pend_off_.clear();
  pend_row_.clear();
}

size_t LabelTable::span_limit() const {
  const size_t rows = std::max<size_t>(row_label_->size() + 1, 1);
  const size_t by_rows = rows * row_bytes_ / 16;  // table bytes <= a quarter of the row matrix
  return std::min(kMaxLimit, std::max(kMinLimit, by_rows));
}

// ---- reads -----------------------------------------------------------------------------------------------------------
void LabelTable::rows_of(uint64_t label, std::vector<uint32_t> &out) const {
  out.clear();
  switch (mode_) {
    case IDENTITY:
      if (label >= identity_base_ && label - identity_base_ < row_label_->size()) out.push_back((uint32_t)(label - identity_base_));
      return;
    case DIRECT: {
      if (label < base_ || label - base_ >= enc_cap_) return;
      uint32_t r = head_at((size_t)(label - base_));
      for (size_t guard = 0; r != kNoRow; r = multi_ ? next_of(r) : kNoRow) {
        out.push_back(r);
        if (++guard > row_label_->size() + 1) throw std::runtime_error("label table: a row chain does not end");
      }
      return;
    }
    case HASH: {
      const size_t sl = hfind(label);
      if (sl == kNoSlot) return;
      uint32_t r = hent_[sl].row;
      for (size_t guard = 0; r != kNoRow; r = multi_ ? next_of(r) : kNoRow) {
        out.push_back(r);
        if (++guard > row_label_->size() + 1) throw std::runtime_error("label table: a row chain does not end");
      }
      return;
    }
  }
}

bool LabelTable::contains(uint64_t label) const {
  switch (mode_) {
    case IDENTITY: return label >= identity_base_ && label - identity_base_ < row_label_->size();
    case DIRECT: return label >= base_ && label - base_ < enc_cap_ && head_at((size_t)(label - base_)) != kNoRow;
    case HASH: {
      const size_t sl = hfind(label);
      return sl != kNoSlot && hent_[sl].row != kNoRow;
    }
  }
  return false;
}

size_t LabelTable::label_count() const {
  switch (mode_) {
    case IDENTITY: return row_label_->size();
    case DIRECT: return n_labels_;
    case HASH: return h_live_;
  }
  return 0;
}

bool LabelTable::any_in_range(uint64_t first, size_t n) const {
  if (!n) return false;
  switch (mode_) {
    case IDENTITY: {
      const uint64_t lo = identity_base_, hi = identity_base_ + row_label_->size();
      return !row_label_->empty() && first < hi && first + n > lo;
    }
    case DIRECT: {
      const uint64_t lo = std::max(first, base_), hi = std::min(first + n, base_ + (uint64_t)enc_cap_);
      for (uint64_t l = lo; l < hi; l++)
        if (head_at((size_t)(l - base_)) != kNoRow) return true;
      return false;
    }
    case HASH:
      for (size_t i = 0; i < n; i++)
        if (contains(first + i)) return true;
      return false;
  }
  return false;
}

// ---- mode changes ----------------------------------------------------------------------------------------------------
// ---- HASH ------------------------------------------------------------------------------------------------------------
void LabelTable::free_hash() {
  hook_free(hent_, hcap_ * sizeof(HEnt));
  hent_ = nullptr;
  hcap_ = h_used_ = h_live_ = 0;
  if (d_hash_) (void)hipFree(d_hash_);
  d_hash_ = nullptr;
  d_hcap_ = 0;
  pend_slot_.clear();
}

size_t LabelTable::hfind(uint64_t label) const {
  if (!hcap_) return kNoSlot;
  const size_t mask = hcap_ - 1;
  const uint32_t lo = (uint32_t)label, hi = (uint32_t)(label >> 32);
  for (size_t p = (size_t)label_hash(label) & mask, g = 0; g < hcap_; g++, p = (p + 1) & mask) {
    const HEnt &e = hent_[p];
    if (!e.used) return kNoSlot;
    if (e.lo == lo && e.hi == hi) return p;
  }
  return kNoSlot;
}

size_t LabelTable::hplace(uint64_t label) {
  const size_t mask = hcap_ - 1;
  const uint32_t lo = (uint32_t)label, hi = (uint32_t)(label >> 32);
  for (size_t p = (size_t)label_hash(label) & mask;; p = (p + 1) & mask) {
    HEnt &e = hent_[p];
    if (!e.used) {
      e = HEnt{lo, hi, kNoRow, 1u};
      h_used_++;
      return p;
    }
    if (e.lo == lo && e.hi == hi) return p;
  }
}

// Every label again, from row_label_ (rows in ascending order: a multi-value label's chain ends up newest first, as the
// incremental inserts build it), into a table of >= 2 x (rows_hint + slack) slots; the device copy is replaced.  Tombstones go.
void LabelTable::hash_rebuild(size_t rows_hint, hipStream_t s) {
  const size_t total = row_label_->size();
  size_t cap = 1024;
  while (cap < 2 * (std::max(rows_hint, total) + 512)) cap <<= 1;
  if (cap > (1ull << 32)) throw std::runtime_error("label table: more labels than a 2^32-slot hash table holds");
  hook_free(hent_, hcap_ * sizeof(HEnt));
  hent_ = static_cast<HEnt *>(hook_calloc(cap, sizeof(HEnt)));
  hcap_ = cap;
  h_used_ = h_live_ = 0;
  pend_slot_.clear();
  pend_row_.clear();
  if (multi_ && nxt_cap_ < total + 1) {
    hook_free(nxt_, nxt_cap_ * sizeof(uint32_t));
    nxt_cap_ = total + 1024;
    nxt_ = static_cast<uint32_t *>(hook_calloc(nxt_cap_, sizeof(uint32_t)));
  }
  for (size_t r = 0; r < total; r++) {
    const size_t sl = hplace((*row_label_)[r]);
    if (hent_[sl].row == kNoRow) h_live_++;
    if (multi_) nxt_[r] = hent_[sl].row == kNoRow ? 0u : hent_[sl].row + 1;
    hent_[sl].row = (uint32_t)r;
  }
  if (d_hash_) HIP_CHECK(hipFree(d_hash_));
  d_hash_ = nullptr;
  d_hcap_ = 0;
  HIP_CHECK(hipMalloc((void **)&d_hash_, cap * sizeof(HEnt)));
  d_hcap_ = cap;
  HIP_CHECK(hipMemcpyAsync(d_hash_, hent_, cap * sizeof(HEnt), hipMemcpyHostToDevice, s));
  if (multi_) {
    ensure_device_next(std::max(row_cap_hint_, total + 1), s);
    if (total) {
      HIP_CHECK(hipMemcpyAsync(d_next_, nxt_, total * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      launch_label_decode(d_next_, total, s);
    }
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));  // (the host table is pageable memory; readers take the new table under the index's lock)
}

// IDENTITY / DIRECT -> HASH: the labels no longer fit a direct table.  O(rows) once, like a rehash; the index stays on the
// device paths (rounds 1-5 went to host hash maps here, and every hybrid entry point with them to its host translation).
void LabelTable::to_hash(hipStream_t s) {
  uint32_t *keep_nxt = nullptr;   // (free_direct drops the chains; hash_rebuild derives them again from row_label_)
  (void)keep_nxt;
  free_direct();
  hash_rebuild(row_label_->size() + 1, s);
  mode_ = HASH;
}

void LabelTable::ensure_device_next(size_t rows, hipStream_t s) {
  if (!multi_ || rows <= d_next_cap_) return;
  const size_t cap = std::max(rows, d_next_cap_ + d_next_cap_ / 2) + 64;
  uint32_t *nn = nullptr;
  HIP_CHECK(hipMalloc((void **)&nn, cap * sizeof(uint32_t)));
  if (d_next_cap_) HIP_CHECK(hipMemcpyAsync(nn, d_next_, d_next_cap_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemsetAsync(nn + d_next_cap_, 0xFF, (cap - d_next_cap_) * sizeof(uint32_t), s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (d_next_) HIP_CHECK(hipFree(d_next_));
  d_next_ = nn;
  d_next_cap_ = cap;
}

// IDENTITY -> DIRECT (or HASH when the labels start too far from zero for a table): every stored row r carries the label
// identity_base_ + r.  Host: a zero page mapping (0 = "as under identity"); device: one fill kernel.
void LabelTable::to_direct(hipStream_t s) {
  const size_t total = row_label_->size();
  base_ = identity_base_ < kSmallBase ? 0 : identity_base_;
  ident_off_ = (size_t)(identity_base_ - base_);
  ident_n_ = total;
  const size_t need = ident_off_ + total, limit = span_limit();
  if (need > limit) {
    to_hash(s);
    return;
  }
  const size_t cap = std::min(limit, need + need / 4 + 1024);
  enc_ = static_cast<uint32_t *>(hook_calloc(cap, sizeof(uint32_t)));
  enc_cap_ = cap;
  HIP_CHECK(hipMalloc((void **)&d_row_of_, cap * sizeof(uint32_t)));
  d_cap_ = cap;
  launch_label_fill(d_row_of_, 0, cap, ident_off_, ident_off_ + ident_n_, 0u, s);
  HIP_CHECK(hipGetLastError());
  n_labels_ = total;
  ensure_device_next(std::max(row_cap_hint_, total + 1), s);
  // (the fill has LANDED before the table is published: on the add() path nothing else synchronises until the next flush, and a
  // reader that took the shared lock right behind this writer would gather from a half-filled table -- round-5 advisor)
  HIP_CHECK(hipStreamSynchronize(s));
  mode_ = DIRECT;
}

void LabelTable::leave_identity(hipStream_t s) {
  if (mode_ == IDENTITY) to_direct(s);
}

// A label below the table's base (rare: doc ids ascend): every entry again, from row_label_, with the new base.
void LabelTable::rebuild_direct(uint64_t new_base, size_t need_span, hipStream_t s) {
  const size_t total = row_label_->size();
  const size_t cap = std::min(span_limit(), need_span + need_span / 4 + 1024);
  free_direct();
  base_ = new_base;
  ident_off_ = ident_n_ = 0;
  enc_ = static_cast<uint32_t *>(hook_calloc(cap, sizeof(uint32_t)));
  enc_cap_ = cap;
  if (multi_) {
    nxt_cap_ = total + 1024;
    nxt_ = static_cast<uint32_t *>(hook_calloc(nxt_cap_, sizeof(uint32_t)));
  }
  n_labels_ = 0;
  for (size_t r = 0; r < total; r++) {
    const size_t off = (size_t)((*row_label_)[r] - base_);
    const uint32_t head = enc_[off];  // 0 = none, row + 1
    if (!head) n_labels_++;
    if (multi_) nxt_[r] = head;
    enc_[off] = (uint32_t)r + 1;
  }
  HIP_CHECK(hipMalloc((void **)&d_row_of_, cap * sizeof(uint32_t)));
  d_cap_ = cap;
  HIP_CHECK(hipMemcpyAsync(d_row_of_, enc_, cap * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  launch_label_decode(d_row_of_, cap, s);
  if (multi_) {
    ensure_device_next(std::max(row_cap_hint_, total + 1), s);
    if (total) {
      HIP_CHECK(hipMemcpyAsync(d_next_, nxt_, total * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      launch_label_decode(d_next_, total, s);
    }
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));
}

void LabelTable::grow_span(size_t need, hipStream_t s) {
  const size_t cap = std::min(span_limit(), need + need / 4 + 1024);
  uint32_t *ne = static_cast<uint32_t *>(hook_calloc(cap, sizeof(uint32_t)));
  memcpy(ne, enc_, enc_cap_ * sizeof(uint32_t));
  hook_free(enc_, enc_cap_ * sizeof(uint32_t));
  enc_ = ne;
  uint32_t *nd = nullptr;
  HIP_CHECK(hipMalloc((void **)&nd, cap * sizeof(uint32_t)));
  HIP_CHECK(hipMemcpyAsync(nd, d_row_of_, d_cap_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  launch_label_fill(nd, d_cap_, cap, 0, 0, 0u, s);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));
  HIP_CHECK(hipFree(d_row_of_));
  d_row_of_ = nd;
  enc_cap_ = d_cap_ = cap;
}

bool LabelTable::slot_for(uint64_t label, size_t n, hipStream_t s, size_t *off) {
  if (label < base_) {
    const uint64_t nb = label < kSmallBase ? 0 : label;
    const uint64_t need = (base_ - nb) + enc_cap_;
    if (need > span_limit() || label + n - nb > span_limit()) {
      to_hash(s);
      return false;
    }
    rebuild_direct(nb, (size_t)std::max<uint64_t>(need, label + n - nb), s);
  }
  const uint64_t o = label - base_;
  if (o + n > enc_cap_) {
    if (o + n > span_limit()) {
      to_hash(s);
      return false;
    }
    grow_span((size_t)(o + n), s);
  }
  *off = (size_t)o;
  return true;
}

// ---- writers ---------------------------------------------------------------------------------------------------------
void LabelTable::set_next(uint32_t row, uint32_t to) {
  if (row >= nxt_cap_) {
    const size_t cap = std::max<size_t>((size_t)row + 1, nxt_cap_ * 2) + 1024;
    uint32_t *nn = static_cast<uint32_t *>(hook_calloc(cap, sizeof(uint32_t)));
    if (nxt_cap_) memcpy(nn, nxt_, nxt_cap_ * sizeof(uint32_t));
    hook_free(nxt_, nxt_cap_ * sizeof(uint32_t));
    nxt_ = nn;
    nxt_cap_ = cap;
  }
  nxt_[row] = to == kNoRow ? 0u : to + 1;
  pend_row_.push_back(row);
}

void LabelTable::insert(uint64_t label, uint32_t row, hipStream_t s) {
  if (mode_ == IDENTITY) {
    if (row == 0) identity_base_ = label;
    if (label == identity_base_ + row) return;
    to_direct(s);  // (row_label_ does not hold `row` yet)
  }
  if (mode_ == DIRECT) {
    size_t off;
    if (slot_for(label, 1, s, &off)) {
      const uint32_t head = head_at(off);
      if (head == kNoRow) n_labels_++;
      if (multi_) set_next(row, head);
      set_head(off, row);
      return;
    }
  }
  // HASH: at most half full, tombstones included
  if (2 * (h_used_ + 1) > hcap_) hash_rebuild(row_label_->size() + 1, s);
  const size_t sl = hplace(label);
  const uint32_t head = hent_[sl].row;
  if (head == kNoRow) h_live_++;
  if (multi_) set_next(row, head);
  hset(sl, row);
}

void LabelTable::insert_range(uint64_t first_label, uint32_t first_row, size_t n, hipStream_t s) {
  if (!n) return;
  if (mode_ == IDENTITY) {
    if (first_row == 0) identity_base_ = first_label;
    if (first_label == identity_base_ + first_row) return;
    to_direct(s);
  }
  if (mode_ == DIRECT) {
    size_t off;
    if (slot_for(first_label, n, s, &off)) {  // (every slot of the range exists now: nothing below grows or rebuilds)
      if (!multi_ || !any_in_range(first_label, n)) {
        for (size_t i = 0; i < n; i++) enc_[off + i] = first_row + (uint32_t)i + 1;
        n_labels_ += n;
        launch_label_fill(d_row_of_, off, off + n, off, off + n, first_row, s);
        if (multi_) {
          for (size_t r = first_row; r < std::min<size_t>((size_t)first_row + n, nxt_cap_); r++) nxt_[r] = 0;
          ensure_device_next(std::max(row_cap_hint_, (size_t)first_row + n), s);
          launch_label_fill(d_next_, first_row, (size_t)first_row + n, 0, 0, 0u, s);
        }
        HIP_CHECK(hipGetLastError());
      } else {
        for (size_t i = 0; i < n; i++) {
          const uint32_t head = head_at(off + i);
          if (head == kNoRow) n_labels_++;
          set_next(first_row + (uint32_t)i, head);
          set_head(off + i, first_row + (uint32_t)i);
        }
      }
      return;
    }
  }
  if (2 * (h_used_ + n) > hcap_) hash_rebuild(row_label_->size() + n, s);
  for (size_t i = 0; i < n; i++) {
    const size_t sl = hplace(first_label + i);
    const uint32_t head = hent_[sl].row;
    if (head == kNoRow) h_live_++;
    if (multi_) set_next(first_row + (uint32_t)i, head);
    hset(sl, first_row + (uint32_t)i);
  }
}

void LabelTable::move_row(uint64_t label, uint32_t from, uint32_t to) {
  if (mode_ == DIRECT) {
    const size_t off = (size_t)(label - base_);
    if (!multi_) {
      set_head(off, to);
      return;
    }
    const uint32_t h = head_at(off);
    if (h == from) {
      set_head(off, to);
    } else {
      uint32_t x = h;
      size_t guard = 0;
      while (x != kNoRow && next_of(x) != from) {
        x = next_of(x);
        if (++guard > row_label_->size() + 1) break;
      }
      if (x == kNoRow || next_of(x) != from) throw std::runtime_error("label table: a moved row is not in its label's chain");
      set_next(x, to);
    }
    set_next(to, next_of(from));
    set_next(from, kNoRow);
    return;
  }
  if (mode_ == HASH) {
    const size_t sl = hfind(label);
    if (sl == kNoSlot) throw std::runtime_error("label table: a moved row's label is not in the table");
    if (!multi_) {
      hset(sl, to);
      return;
    }
    const uint32_t h = hent_[sl].row;
    if (h == from) {
      hset(sl, to);
    } else {
      uint32_t x = h;
      size_t guard = 0;
      while (x != kNoRow && next_of(x) != from) {
        x = next_of(x);
        if (++guard > row_label_->size() + 1) break;
      }
      if (x == kNoRow || next_of(x) != from) throw std::runtime_error("label table: a moved row is not in its label's chain");
      set_next(x, to);
    }
    set_next(to, next_of(from));
    set_next(from, kNoRow);
  }
}

void LabelTable::erase_label(uint64_t label) {
  if (mode_ == DIRECT) {
    if (label < base_ || label - base_ >= enc_cap_) return;
    const size_t off = (size_t)(label - base_);
    if (head_at(off) != kNoRow) n_labels_--;
    set_head(off, kNoRow);
    return;
  }
  if (mode_ == HASH) {
    const size_t sl = hfind(label);
    if (sl == kNoSlot) return;
    if (hent_[sl].row != kNoRow) h_live_--;
    hset(sl, kNoRow);  // a tombstone: the label keeps its slot (labels are never reused), the next rebuild drops it
  }
}

void LabelTable::sync_device(hipStream_t s) {
  if (mode_ == IDENTITY) {
    pend_off_.clear();
    pend_row_.clear();
    pend_slot_.clear();
    return;
  }
  const bool hash = mode_ == HASH;
  const size_t n1 = hash ? pend_slot_.size() : pend_off_.size(), n2 = multi_ ? pend_row_.size() : 0;
  if (multi_) ensure_device_next(std::max(row_cap_hint_, row_label_->size() + 1), s);
  if (!(n1 + n2)) {
    pend_row_.clear();
    return;
  }
  // pinned block: [n1 indices][n1 values of 1 (direct) or 4 (hash) words, 16-byte aligned][n2 rows][n2 next values]
  const size_t vw = hash ? 4 : 1, v1_at = (n1 + 3) / 4 * 4, i2_at = v1_at + vw * n1, words = i2_at + 2 * n2;
  if (words > pin_cap_) {
    if (h_pin_) HIP_CHECK(hipHostFree(h_pin_));
    if (d_pend_) HIP_CHECK(hipFree(d_pend_));
    h_pin_ = d_pend_ = nullptr;
    pin_cap_ = 0;
    const size_t cap = std::max<size_t>(words, 16384);
    HIP_CHECK(hipHostMalloc((void **)&h_pin_, cap * sizeof(uint32_t), hipHostMallocDefault));
    HIP_CHECK(hipMalloc((void **)&d_pend_, cap * sizeof(uint32_t)));
    pin_cap_ = cap;
  }
  // the values are read HERE, from the host table: several updates of one entry all carry its final value
  uint32_t *i1 = h_pin_, *v1 = h_pin_ + v1_at, *i2 = h_pin_ + i2_at, *v2 = i2 + n2;
  for (size_t i = 0; i < n1; i++) {
    if (hash) {
      i1[i] = pend_slot_[i];
      memcpy(v1 + 4 * i, &hent_[pend_slot_[i]], sizeof(HEnt));
    } else {
      i1[i] = pend_off_[i];
      v1[i] = head_at(pend_off_[i]);
    }
  }
  for (size_t i = 0; i < n2; i++) {
    i2[i] = pend_row_[i];
    v2[i] = next_of(pend_row_[i]);
  }
  HIP_CHECK(hipMemcpyAsync(d_pend_, h_pin_, words * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  if (hash) launch_label_scatter16(d_hash_, d_pend_, d_pend_ + v1_at, (uint32_t)n1, s);
  else launch_label_scatter(d_row_of_, d_pend_, d_pend_ + v1_at, (uint32_t)n1, s);
  if (n2) launch_label_scatter(d_next_, d_pend_ + i2_at, d_pend_ + i2_at + n2, (uint32_t)n2, s);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));  // (the pinned block is reused by the next call)
  pend_off_.clear();
  pend_row_.clear();
  pend_slot_.clear();
}

bool LabelTable::device_view(uint32_t committed_rows, LabelRows *out) const {
  if (mode_ == HASH) {
    *out = LabelRows{nullptr, multi_ ? d_next_ : nullptr, 0, 0, committed_rows, d_hash_, (uint32_t)(d_hcap_ - 1)};
    return true;
  }
  if (mode_ == IDENTITY) {
    *out = LabelRows{nullptr, nullptr, identity_base_, committed_rows, committed_rows, nullptr, 0};
    return true;
  }
  *out = LabelRows{d_row_of_, multi_ ? d_next_ : nullptr, base_, (uint32_t)std::min<size_t>(d_cap_, 0xFFFFFFFFu), committed_rows, nullptr, 0};
  return true;
}

}  // namespace rsgpu
