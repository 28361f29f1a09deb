// label_table.hpp -- label (doc id) -> storage row(s) of a FLAT index: a host copy for the writers and a device copy the
// kernels read.
//
// The reference looks vectors up by label on every ad-hoc candidate (src/iterators/hybrid_reader.c:309-327 ->
// VecSimIndex_GetDistanceFrom_Unsafe), and its labels are doc ids: dense, ascending, never reused; documents without the
// vector field have no row, an update is delete + a NEW id (src/indexer.c:179-190), deletes arrive through
// VecSimIndex_DeleteVector (src/spec.c:3533-3541).  Three forms, chosen by what the labels look like:
//
//   IDENTITY  label == base + row for every row (a bulk-loaded or append-only index): no table at all.
//   DIRECT    a direct-addressed table over [base, base + span): row_of[label - base] = (first) row or "none"; multi-value
//             indexes chain a label's rows through next[row].  One u32 per label in HBM (200 MB for 50 M doc ids) -- the
//             tile kernels, labels_to_rows and the ad-hoc gather translate on the device, and AddVector / DeleteVector touch
//             two or three entries.  Leaving IDENTITY is O(1) on the host: the host table is calloc'ed and encodes
//             "unchanged since identity" as 0, so no page is touched until a label changes; the device table is one fill
//             kernel.  Allowed while the table stays below max(4 MiB, 1/4 of the row matrix).
//   HASH      (round 6; replaces the host hash maps of rounds 1-5) labels too far apart for that -- in the reference a label is
//             a doc id of the WHOLE document table (src/document.c:712-725: only documents with the vector field get a row),
//             so "1 M vectors in a 10^9-document index" is the ordinary case, and a long-lived index's span only grows
//             (src/indexer.c:179-190) -- an open-addressing table in HBM with a host copy: 16-byte slots {label, first row,
//             used}, linear probing from a splitmix hash, at most half full (32 bytes per row), deleted labels stay as
//             tombstones (labels are never reused) until a rebuild.  The same kernels read it (kernels.hpp label_first_row):
//             the hybrid entry points keep their tile paths whatever the labels look like.
#pragma once
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace rsgpu {

class LabelTable {
 public:
  enum Mode { IDENTITY = 0, DIRECT = 1, HASH = 2 };
  using LabelVec = std::vector<uint64_t, HookAlloc<uint64_t>>;
  // row_label: the index's row -> label vector (committed + staged rows); an insert of row r is announced BEFORE r is
  // appended to it.  host_bytes: the index's counter of host memory taken through the installed memory functions.
  LabelTable(bool multi, size_t row_bytes, const LabelVec *row_label, size_t *host_bytes);
  ~LabelTable();
  LabelTable(const LabelTable &) = delete;
  LabelTable &operator=(const LabelTable &) = delete;

  Mode mode() const { return mode_; }
  bool identity(uint64_t *base) const {
    if (base) *base = identity_base_;
    return mode_ == IDENTITY;
  }
  // rows (committed and staged) stored under `label`; rows = row_label->size()
  void rows_of(uint64_t label, std::vector<uint32_t> &out) const;
  bool contains(uint64_t label) const;
  size_t label_count() const;  // distinct labels
  // any label of [first, first + n) stored?
  bool any_in_range(uint64_t first, size_t n) const;

  // ---- writers (the index's unique lock is held; device work goes to `s`, see sync_device) ----
  void insert(uint64_t label, uint32_t row, hipStream_t s);
  void insert_range(uint64_t first_label, uint32_t first_row, size_t n, hipStream_t s);
  // the first delete: identity labelling ends (the table appears), nothing else changes
  void leave_identity(hipStream_t s);
  // row `from` (the last row) of `label` now lives at `to` (a row just freed)
  void move_row(uint64_t label, uint32_t from, uint32_t to);
  void erase_label(uint64_t label);
  // device rows the chains must cover (the index's row capacity)
  void set_row_capacity(size_t rows) { row_cap_hint_ = rows; }
  // queue the pending entry updates behind the stream's work; the caller synchronises `s` before it lets readers in
  void sync_device(hipStream_t s);

  // what the kernels take (every mode has a device form since round 6: always true)
  bool device_view(uint32_t committed_rows, LabelRows *out) const;
  size_t device_bytes() const { return (d_cap_ + d_next_cap_) * sizeof(uint32_t) + d_hcap_ * sizeof(HEnt); }

 private:
  static constexpr uint32_t kTomb = 0xFFFFFFFFu;  // host encoding: 0 = unchanged since identity, row + 1, kTomb = deleted
  size_t span_limit() const;
  uint32_t head_at(size_t off) const {
    const uint32_t v = enc_[off];
    if (v == 0) return (off >= ident_off_ && off - ident_off_ < ident_n_) ? (uint32_t)(off - ident_off_) : kNoRow;
    return v == kTomb ? kNoRow : v - 1;
  }
  uint32_t next_of(uint32_t row) const { return (row < nxt_cap_ && nxt_[row]) ? nxt_[row] - 1 : kNoRow; }
  void set_head(size_t off, uint32_t row) {
    enc_[off] = row == kNoRow ? kTomb : row + 1;
    pend_off_.push_back((uint32_t)off);
  }
  void set_next(uint32_t row, uint32_t to);
  void to_direct(hipStream_t s);
  void to_hash(hipStream_t s);
  void rebuild_direct(uint64_t new_base, size_t need_span, hipStream_t s);
  // the slot of `label`, growing / rebasing the table; false: the table went HASH
  bool slot_for(uint64_t label, size_t n, hipStream_t s, size_t *off);
  void grow_span(size_t need, hipStream_t s);
  void ensure_device_next(size_t rows, hipStream_t s);
  void free_direct();
  void *hook_calloc(size_t n, size_t sz);
  void hook_free(void *p, size_t bytes);

  const bool multi_;
  const size_t row_bytes_;
  const LabelVec *row_label_;
  size_t *host_bytes_;
  Mode mode_ = IDENTITY;
  uint64_t identity_base_ = 0;
  size_t n_labels_ = 0;  // DIRECT: distinct labels
  // DIRECT, host
  uint64_t base_ = 0;
  uint32_t *enc_ = nullptr;
  size_t enc_cap_ = 0;
  size_t ident_off_ = 0, ident_n_ = 0;
  uint32_t *nxt_ = nullptr;  // multi: 0 = none, row + 1
  size_t nxt_cap_ = 0;
  // DIRECT, device
  uint32_t *d_row_of_ = nullptr;
  size_t d_cap_ = 0;
  uint32_t *d_next_ = nullptr;
  size_t d_next_cap_ = 0, row_cap_hint_ = 0;
  std::vector<uint32_t> pend_off_, pend_row_;
  uint32_t *h_pin_ = nullptr, *d_pend_ = nullptr;
  size_t pin_cap_ = 0;
  // HASH: host copy + device copy of the open-addressing table
  struct HEnt {
    uint32_t lo, hi, row, used;  // label, first row (kNoRow: deleted), 0 = empty slot
  };
  static constexpr size_t kNoSlot = ~(size_t)0;
  HEnt *hent_ = nullptr;
  size_t hcap_ = 0, h_used_ = 0, h_live_ = 0;  // slots (a power of two), used slots (live + tombstones), live labels
  HEnt *d_hash_ = nullptr;
  size_t d_hcap_ = 0;
  std::vector<uint32_t> pend_slot_;
  size_t hfind(uint64_t label) const;                       // the label's slot (live or tombstone) or kNoSlot
  size_t hplace(uint64_t label);                            // ... or a fresh slot for it (the table has room)
  void hset(size_t slot, uint32_t row) {
    hent_[slot].row = row;
    pend_slot_.push_back((uint32_t)slot);
  }
  void hash_rebuild(size_t rows_hint, hipStream_t s);       // every label again, from row_label_ (+ chains), uploaded
  void free_hash();
};

}  // namespace rsgpu
