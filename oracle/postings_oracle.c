/*
 * oracle/postings_oracle.c -- CPU restatement of RediSearch's posting-list codecs, block reader and
 * N-way intersection.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/flat_oracle.c header for the rule).
 *
 * Follows:
 *   reference src/redisearch_rs/qint/src/lib.rs:49-214        qint_encode / qint_decode
 *   reference src/redisearch_rs/varint/src/lib.rs             write_as_varint / read_as_varint
 *   reference src/redisearch_rs/inverted_index/src/index/core.rs:76-96,180-330
 *                                                            IndexBlock, add_entry, take_block
 *   reference src/redisearch_rs/inverted_index/src/codec/{full,freqs_fields,freqs_only,
 *             fields_only,fields_offsets,offsets_only,freqs_offsets,doc_ids_only,
 *             raw_doc_ids_only}.rs                           record layouts, block sizes
 *   reference src/redisearch_rs/inverted_index/src/reader/core.rs  next_record / seek_record / skip_to
 *   reference src/redisearch_rs/rqe_iterators/src/intersection.rs:60-119,256-288,428-452
 *                                                            child ordering, find_consensus, read
 * All integer work: results must be bit-identical to the reference's.
 * Pinned by tests/test_oracle_postings.py against the reference's byte-exact codec tests and
 * tests/cpptests/test_cpp_index.cpp:542-601.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

enum {
  C_FULL = 0, C_FREQS_FIELDS = 1, C_FREQS_ONLY = 2, C_FIELDS_ONLY = 3, C_FIELDS_OFFSETS = 4,
  C_OFFSETS_ONLY = 5, C_FREQS_OFFSETS = 6, C_DOCIDS_ONLY = 7, C_RAW_DOCIDS = 8,
  /* wide field masks (u128 as a varint behind the qint fields / the varint delta): reference codec/full.rs:181-213,
   * freqs_fields.rs:100-140, fields_only.rs:95-130, fields_offsets.rs:125-175 */
  C_FULL_WIDE = 9, C_FREQS_FIELDS_WIDE = 10, C_FIELDS_ONLY_WIDE = 11, C_FIELDS_OFFSETS_WIDE = 12, C_NUM = 13
};
/* qint arity and which slot holds what (-1 = absent). slot 0 is always the delta. wide: a varint u128 mask follows;
 * n == 0 with wide: the delta is a varint too (FieldsOnlyWide). */
typedef struct { int n, freq, mask, osz; uint16_t block_entries; int wide; } CodecDesc;
static const CodecDesc CODECS[C_NUM] = {
  {4, 1, 2, 3, 100, 0}, {3, 1, 2, -1, 100, 0}, {2, 1, -1, -1, 100, 0}, {2, -1, 1, -1, 100, 0}, {3, -1, 1, 2, 100, 0},
  {2, -1, -1, 1, 100, 0}, {3, 1, -1, 2, 100, 0}, {0, -1, -1, -1, 1000, 0}, {0, -1, -1, -1, 1000, 0},
  {3, 1, -1, 2, 100, 1}, {2, 1, -1, -1, 100, 1}, {0, -1, -1, -1, 100, 1}, {2, -1, -1, 1, 100, 1},
};
typedef unsigned __int128 u128;

/* ---- qint ---------------------------------------------------------------------------------------- */
/* Encodes n (2..4) u32 values; returns bytes written. Header byte: 2 bits per value = len-1. */
size_t oracle_qint_encode(uint8_t *out, const uint32_t *vals, int n) {
  uint8_t leading = 0; size_t pos = 1;
  for (int i = 0; i < n; i++) {
    uint32_t v = vals[i]; int len = 0;
    do { out[pos++] = (uint8_t)v; v >>= 8; len++; } while (v);
    leading |= (uint8_t)((len - 1) << (i * 2));
  }
  out[0] = leading;
  return pos;
}
/* Returns bytes consumed, or 0 if the buffer is too short (UnexpectedEof upstream). */
size_t oracle_qint_decode(const uint8_t *in, size_t avail, uint32_t *vals, int n) {
  if (avail < 1) return 0;
  uint8_t leading = in[0]; size_t pos = 1;
  for (int i = 0; i < n; i++) {
    int len = ((leading >> (i * 2)) & 3) + 1;
    if (pos + (size_t)len > avail) return 0;
    uint32_t v = 0;
    for (int b = 0; b < len; b++) v |= (uint32_t)in[pos + b] << (8 * b);
    vals[i] = v; pos += (size_t)len;
  }
  return pos;
}

/* ---- varint (big-endian groups of 7 with the "minus one per continuation" trick) ------------------ */
size_t oracle_varint_encode(uint8_t *out, uint64_t v) {
  uint8_t buf[16]; int pos = 15;
  buf[pos] = (uint8_t)(v & 0x7f); v >>= 7;
  while (v) { v--; buf[--pos] = (uint8_t)(0x80 | (v & 0x7f)); v >>= 7; }
  memcpy(out, buf + pos, (size_t)(16 - pos));
  return (size_t)(16 - pos);
}
size_t oracle_varint_decode(const uint8_t *in, size_t avail, uint64_t *v) {
  if (!avail) return 0;
  size_t pos = 0; uint8_t c = in[pos++]; uint64_t val = c & 0x7f;
  while (c & 0x80) {
    if (pos >= avail) return 0;
    val++; c = in[pos++]; val = (val << 7) | (c & 0x7f);
  }
  *v = val; return pos;
}

size_t oracle_varint128_encode(uint8_t *out, uint64_t lo, uint64_t hi) {
  u128 v = ((u128)hi << 64) | lo;
  uint8_t buf[24]; int pos = 23;
  buf[pos] = (uint8_t)(v & 0x7f); v >>= 7;
  while (v) { v--; buf[--pos] = (uint8_t)(0x80 | (uint8_t)(v & 0x7f)); v >>= 7; }
  memcpy(out, buf + pos, (size_t)(24 - pos));
  return (size_t)(24 - pos);
}
size_t oracle_varint128_decode(const uint8_t *in, size_t avail, uint64_t *lo, uint64_t *hi) {
  if (!avail) return 0;
  size_t pos = 0; uint8_t c = in[pos++]; u128 val = c & 0x7f;
  while (c & 0x80) {
    if (pos >= avail) return 0;
    val++; c = in[pos++]; val = (val << 7) | (c & 0x7f);
  }
  *lo = (uint64_t)val; *hi = (uint64_t)(val >> 64); return pos;
}

/* ---- inverted index ------------------------------------------------------------------------------ */
typedef struct { uint64_t first, last; uint16_t n; uint8_t *buf; size_t len, cap; } OBlock;
typedef struct { int codec; OBlock *b; size_t nb, capb; uint32_t n_unique; } OInv;

OInv *oinv_new(int codec) {
  if (codec < 0 || codec >= C_NUM) return NULL;
  OInv *ii = calloc(1, sizeof *ii); ii->codec = codec; return ii;
}
void oinv_free(OInv *ii) {
  if (!ii) return;
  for (size_t i = 0; i < ii->nb; i++) free(ii->b[i].buf);
  free(ii->b); free(ii);
}
static OBlock *new_block(OInv *ii, uint64_t doc) {
  if (ii->nb == ii->capb) { ii->capb = ii->capb ? ii->capb * 2 : 4; ii->b = realloc(ii->b, ii->capb * sizeof(OBlock)); }
  OBlock *bl = &ii->b[ii->nb++];
  memset(bl, 0, sizeof *bl); bl->first = bl->last = doc;
  return bl;
}
/* add_entry: same-doc duplicates are skipped; a full block or a >u32 delta opens a new block whose
 * first entry has delta 0. Returns 1 if a record was written. */
int oinv_add_wide(OInv *ii, uint64_t doc, uint32_t freq, uint64_t mask_lo, uint64_t mask_hi, const uint8_t *offs, uint32_t osz);
int oinv_add(OInv *ii, uint64_t doc, uint32_t freq, uint32_t mask, const uint8_t *offs, uint32_t osz) {
  return oinv_add_wide(ii, doc, freq, mask, 0, offs, osz);
}
int oinv_add_wide(OInv *ii, uint64_t doc, uint32_t freq, uint64_t mask_lo, uint64_t mask_hi, const uint8_t *offs, uint32_t osz) {
  const CodecDesc *cd = &CODECS[ii->codec];
  const uint32_t mask = (uint32_t)mask_lo;
  if (ii->nb && ii->b[ii->nb - 1].last == doc && ii->b[ii->nb - 1].n) return 0;
  OBlock *bl = (ii->nb && ii->b[ii->nb - 1].n < cd->block_entries) ? &ii->b[ii->nb - 1] : new_block(ii, doc);
  uint64_t base = (ii->codec == C_RAW_DOCIDS) ? bl->first : bl->last;
  uint64_t delta = doc - base;
  if (delta > 0xFFFFFFFFull) { bl = new_block(ii, doc); delta = 0; }
  size_t need = bl->len + 64 + (cd->osz >= 0 ? osz : 0);
  if (need > bl->cap) { bl->cap = need * 2; bl->buf = realloc(bl->buf, bl->cap); }
  uint8_t *w = bl->buf + bl->len;
  if (ii->codec == C_DOCIDS_ONLY) bl->len += oracle_varint_encode(w, delta);
  else if (ii->codec == C_RAW_DOCIDS) { uint32_t d = (uint32_t)delta; memcpy(w, &d, 4); bl->len += 4; }
  else if (ii->codec == C_FIELDS_ONLY_WIDE) {
    size_t k = oracle_varint_encode(w, delta);
    k += oracle_varint128_encode(w + k, mask_lo, mask_hi);
    bl->len += k;
  } else {
    uint32_t v[4]; v[0] = (uint32_t)delta;
    if (cd->freq >= 0) v[cd->freq] = freq;
    if (cd->mask >= 0) v[cd->mask] = mask;
    if (cd->osz >= 0) v[cd->osz] = osz;
    size_t k = oracle_qint_encode(w, v, cd->n);
    if (cd->wide) k += oracle_varint128_encode(w + k, mask_lo, mask_hi);
    if (cd->osz >= 0 && osz) { memcpy(w + k, offs, osz); k += osz; }
    bl->len += k;
  }
  bl->n++; bl->last = doc; ii->n_unique++;
  return 1;
}
void oinv_add_many(OInv *ii, const uint64_t *docs, const uint32_t *freqs, size_t n) {
  for (size_t i = 0; i < n; i++) oinv_add(ii, docs[i], freqs ? freqs[i] : 1, 1, NULL, 0);
}
size_t oinv_num_blocks(const OInv *ii) { return ii->nb; }
uint32_t oinv_unique_docs(const OInv *ii) { return ii->n_unique; }
size_t oinv_total_bytes(const OInv *ii) { size_t s = 0; for (size_t i = 0; i < ii->nb; i++) s += ii->b[i].len; return s; }
/* Flatten to the upload format: per-block header arrays + one concatenated byte buffer. */
void oinv_flatten(const OInv *ii, uint64_t *first, uint64_t *last, uint32_t *nent, uint64_t *off, uint8_t *bytes) {
  size_t o = 0;
  for (size_t i = 0; i < ii->nb; i++) {
    first[i] = ii->b[i].first; last[i] = ii->b[i].last; nent[i] = ii->b[i].n; off[i] = o;
    memcpy(bytes + o, ii->b[i].buf, ii->b[i].len); o += ii->b[i].len;
  }
  off[ii->nb] = o;
}

/* ---- reader --------------------------------------------------------------------------------------- */
typedef struct {
  const OInv *ii; size_t blk; size_t pos; uint64_t last_doc;
  /* current record */
  uint64_t doc; uint32_t freq, mask; const uint8_t *offs; uint32_t osz;
  uint64_t mask_lo, mask_hi;
  int eof;
} OReader;

static void set_block(OReader *r, size_t i) { r->blk = i; r->pos = 0; r->last_doc = r->ii->b[i].first; }
OReader *oreader_new(const OInv *ii) {
  OReader *r = calloc(1, sizeof *r); r->ii = ii;
  if (ii->nb) set_block(r, 0);
  return r;
}
void oreader_free(OReader *r) { free(r); }
void oreader_rewind(OReader *r) { r->eof = 0; r->doc = 0; if (r->ii->nb) set_block(r, 0); else { r->pos = 0; r->last_doc = 0; } }

/* decode one record at the cursor; returns 0 on end-of-buffer */
static int decode_one(OReader *r, uint64_t base) {
  const OBlock *bl = &r->ii->b[r->blk]; const CodecDesc *cd = &CODECS[r->ii->codec];
  const uint8_t *p = bl->buf + r->pos; size_t avail = bl->len - r->pos;
  /* a record keeps what its codec does not store from the reader's base result: frequency 1
   * (RawTermResultBuilder::new, index_result/src/core/mod.rs:192-197; rqe_iterators/src/inverted_index/term.rs:93) */
  r->freq = cd->freq >= 0 ? 0 : 1; r->mask = 0; r->offs = NULL; r->osz = 0; r->mask_lo = r->mask_hi = 0;
  if (r->ii->codec == C_FIELDS_ONLY_WIDE) {
    uint64_t d; size_t k = oracle_varint_decode(p, avail, &d); if (!k) return 0;
    size_t k2 = oracle_varint128_decode(p + k, avail - k, &r->mask_lo, &r->mask_hi); if (!k2) return 0;
    r->doc = base + (uint32_t)d; r->mask = (uint32_t)r->mask_lo; r->pos += k + k2; return 1;
  }
  if (r->ii->codec == C_DOCIDS_ONLY) {
    uint64_t d; size_t k = oracle_varint_decode(p, avail, &d); if (!k) return 0;
    r->doc = base + (uint32_t)d; r->pos += k; return 1;
  }
  if (r->ii->codec == C_RAW_DOCIDS) {
    if (avail < 4) return 0;
    uint32_t d; memcpy(&d, p, 4); r->doc = bl->first + d; r->pos += 4; return 1;
  }
  uint32_t v[4]; size_t k = oracle_qint_decode(p, avail, v, cd->n); if (!k) return 0;
  uint32_t osz = cd->osz >= 0 ? v[cd->osz] : 0;
  if (cd->wide) {
    size_t k2 = oracle_varint128_decode(p + k, avail - k, &r->mask_lo, &r->mask_hi); if (!k2) return 0;
    k += k2; r->mask = (uint32_t)r->mask_lo;
  }
  if (k + osz > avail) return 0;
  r->doc = base + v[0];
  if (cd->freq >= 0) r->freq = v[cd->freq];
  if (cd->mask >= 0) { r->mask = v[cd->mask]; r->mask_lo = r->mask; }
  if (cd->osz >= 0) { r->offs = p + k; r->osz = osz; }
  r->pos += k + osz;
  return 1;
}
/* next_record: 1 = record available, 0 = EOF */
int oreader_next(OReader *r) {
  if (r->eof || !r->ii->nb) { r->eof = 1; return 0; }
  if (r->ii->b[r->blk].len <= r->pos) {
    if (r->blk + 1 >= r->ii->nb) { r->eof = 1; return 0; }
    set_block(r, r->blk + 1);
  }
  if (!decode_one(r, r->last_doc)) { r->eof = 1; return 0; }
  r->last_doc = r->doc;
  return 1;
}
/* block-level skip: next-block shortcut, then binary search on last_doc_id */
static int skip_block(OReader *r, uint64_t target) {
  const OInv *ii = r->ii;
  if (!ii->nb) return 0;
  if (ii->b[r->blk].last >= target) return 1;
  if (ii->b[ii->nb - 1].last < target) return 0;
  size_t s = r->blk + 1;
  if (s < ii->nb && ii->b[s].last >= target) { set_block(r, s); return 1; }
  size_t lo = s, hi = ii->nb;
  while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (ii->b[mid].last < target) lo = mid + 1; else hi = mid; }
  set_block(r, lo);
  return 1;
}
/* seek_record: first record with doc >= target. 1 = positioned (r->doc >= target), 0 = EOF */
int oreader_seek(OReader *r, uint64_t target) {
  if (r->eof) return 0;
  if (!skip_block(r, target)) { r->eof = 1; return 0; }
  uint64_t base = r->last_doc;
  for (;;) {
    if (!decode_one(r, base)) { r->eof = 1; return 0; }
    base = r->doc;
    if (r->doc >= target) break;
  }
  r->last_doc = r->doc;
  return 1;
}
uint64_t oreader_doc(const OReader *r) { return r->doc; }
uint32_t oreader_freq(const OReader *r) { return r->freq; }
uint32_t oreader_mask(const OReader *r) { return r->mask; }
void oreader_mask128(const OReader *r, uint64_t *lo, uint64_t *hi) { *lo = r->mask_lo; *hi = r->mask_hi; }
/* all records' 128-bit masks (wide codecs) */
size_t oinv_decode_masks128(const OInv *ii, uint64_t *lo, uint64_t *hi);
uint32_t oreader_offsets(const OReader *r, const uint8_t **p) { *p = r->offs; return r->osz; }

/* decode everything (ids/freqs/masks may be NULL); returns the number of records */
size_t oinv_decode_all(const OInv *ii, uint64_t *ids, uint32_t *freqs, uint32_t *masks) {
  OReader *r = oreader_new(ii); size_t n = 0;
  while (oreader_next(r)) {
    if (ids) ids[n] = r->doc; if (freqs) freqs[n] = r->freq; if (masks) masks[n] = r->mask;
    n++;
  }
  oreader_free(r);
  return n;
}

size_t oinv_decode_masks128(const OInv *ii, uint64_t *lo, uint64_t *hi) {
  OReader *r = oreader_new(ii); size_t n = 0;
  while (oreader_next(r)) { lo[n] = r->mask_lo; hi[n] = r->mask_hi; n++; }
  oreader_free(r);
  return n;
}

/* ---- N-way intersection ---------------------------------------------------------------------------
 * Children are ordered by estimated size (unique docs) ascending, stably; the first child drives,
 * every other child is skipped to the candidate; a child landing past it restarts the round with
 * its doc id (find_consensus).  Outputs: hit doc ids, and per ORIGINAL list index the matched
 * freq / field mask, laid out [list][hit] with stride `cap`.  Returns the number of hits (stops
 * at cap). */
typedef struct { OReader *r; size_t orig; uint64_t cur; } Child;
size_t oracle_intersect(const OInv **lists, size_t nl, size_t cap, uint64_t *ids, uint32_t *freqs, uint32_t *masks) {
  if (!nl) return 0;
  Child *c = malloc(nl * sizeof *c);
  for (size_t i = 0; i < nl; i++) { c[i].r = oreader_new(lists[i]); c[i].orig = i; c[i].cur = 0; }
  for (size_t i = 1; i < nl; i++) { /* stable insertion sort by num_estimated */
    Child t = c[i]; size_t j = i;
    while (j && c[j - 1].r->ii->n_unique > t.r->ii->n_unique) { c[j] = c[j - 1]; j--; }
    c[j] = t;
  }
  size_t hits = 0; int eof = 0;
  while (!eof && hits < cap) {
    if (!oreader_next(c[0].r)) break;
    uint64_t target = c[0].cur = c[0].r->doc;
    for (;;) { /* find_consensus */
      int agreed = 1;
      for (size_t i = 0; i < nl; i++) {
        if (c[i].cur == target) continue;
        if (!oreader_seek(c[i].r, target)) { eof = 1; agreed = 0; break; }
        c[i].cur = c[i].r->doc;
        if (c[i].cur != target) { target = c[i].cur; agreed = 0; break; }
      }
      if (eof || agreed) break;
    }
    if (eof) break;
    ids[hits] = target;
    for (size_t i = 0; i < nl; i++) {
      if (freqs) freqs[c[i].orig * cap + hits] = c[i].r->freq;
      if (masks) masks[c[i].orig * cap + hits] = c[i].r->mask;
    }
    hits++;
  }
  for (size_t i = 0; i < nl; i++) oreader_free(c[i].r);
  free(c);
  return hits;
}

/* ---- proximity: max_slop / in_order (reference src/redisearch_rs/index_result/src/core/proximity.rs:134-298) and the
 * slop a scorer divides by (reference src/index_result/index_result.c:51-103 IndexResult_MinOffsetDelta), over the
 * varint-delta offset bytes of the children's current records.  A child is one term (`n_leaves` = 1) or a union /
 * intersection of terms whose positions are merged in ascending order (proximity.rs OffsetIter::Merge). -------------- */
#define P_EOF 0xFFFFFFFFu
typedef struct { const uint8_t *p; size_t len, pos; uint32_t last; } PTerm;
static uint32_t pterm_next(PTerm *t) {
  if (t->pos >= t->len) return P_EOF;
  uint64_t d; size_t k = oracle_varint_decode(t->p + t->pos, t->len - t->pos, &d);
  if (!k) { t->pos = t->len; return P_EOF; }
  t->pos += k; t->last += (uint32_t)d;
  return t->last;
}
typedef struct { PTerm *leaf; uint32_t *look; size_t n; int merged; } PChild;
static void pchild_prime(PChild *c) { if (c->merged) for (size_t i = 0; i < c->n; i++) c->look[i] = pterm_next(&c->leaf[i]); }
static uint32_t pchild_next(PChild *c) {
  if (!c->merged) return c->n ? pterm_next(&c->leaf[0]) : P_EOF;
  size_t best = c->n; uint32_t mv = P_EOF;
  for (size_t i = 0; i < c->n; i++) if (c->look[i] != P_EOF && c->look[i] < mv) { mv = c->look[i]; best = i; }
  if (best == c->n) return P_EOF;
  c->look[best] = pterm_next(&c->leaf[best]);
  return mv;
}
static int prox_in_order(PChild *it, size_t n, uint32_t max_slop) {
  uint32_t *positions = calloc(n, sizeof *positions);
  int result = -1;
  while (result < 0) {
    int32_t span = 0; int over = 0;
    for (size_t i = 0; i < n; i++) {
      uint32_t pos;
      if (i == 0) { pos = pchild_next(&it[0]); if (pos == P_EOF) { result = 0; break; } }
      else pos = positions[i];
      uint32_t last_pos = i == 0 ? 0u : positions[i - 1];
      while (pos < last_pos) { pos = pchild_next(&it[i]); if (pos == P_EOF) { result = 0; break; } }
      if (result == 0) break;
      positions[i] = pos;
      if (i > 0) {
        span += (int32_t)pos - (int32_t)last_pos - 1;
        if (span > 0 && (uint32_t)span > max_slop) { over = 1; break; }
      }
    }
    if (result < 0 && !over) result = 1;
  }
  free(positions);
  return result;
}
static int prox_unordered(PChild *it, size_t n, uint32_t max_slop) {
  uint32_t *positions = calloc(n, sizeof *positions);
  for (size_t i = 0; i < n; i++) { positions[i] = pchild_next(&it[i]); if (positions[i] == P_EOF) { free(positions); return 0; } }
  uint32_t max_pos = 0;
  for (size_t i = 0; i < n; i++) if (positions[i] >= max_pos) max_pos = positions[i];
  int result = 0;
  for (;;) {
    uint32_t min_pos = P_EOF; size_t min_idx = 0;
    for (size_t i = 0; i < n; i++) if (positions[i] < min_pos) { min_pos = positions[i]; min_idx = i; }
    if (min_pos != max_pos) {
      int32_t span = (int32_t)max_pos - (int32_t)min_pos - ((int32_t)n - 1);
      if (span < 0 || (uint32_t)span <= max_slop) { result = 1; break; }
    }
    uint32_t np = pchild_next(&it[min_idx]);
    if (np == P_EOF) break;
    positions[min_idx] = np;
    if (np > max_pos) max_pos = np;
  }
  free(positions);
  return result;
}
/* children in aggregate order: child c owns leaves [child_first[c], child_first[c+1]); is_agg[c] != 0 marks a union /
 * intersection child (it "has offsets" whatever its leaves hold, proximity.rs:72-90).  off/len: the offset bytes of
 * every leaf's current record (len 0 = none / leaf absent).  max_slop < 0: no slop constraint. */
int oracle_within_range(size_t n_children, const size_t *child_first, const int *is_agg, const uint8_t *const *off,
                        const uint32_t *len, long max_slop, int in_order) {
  if (n_children <= 1) return 1;
  size_t n_leaves = child_first[n_children];
  PTerm *leaf = calloc(n_leaves ? n_leaves : 1, sizeof *leaf);
  uint32_t *look = calloc(n_leaves ? n_leaves : 1, sizeof *look);
  PChild *it = calloc(n_children, sizeof *it);
  size_t m = 0;
  for (size_t c = 0; c < n_children; c++) {
    size_t a = child_first[c], b = child_first[c + 1];
    int has = is_agg[c] ? 1 : (b > a && len[a] > 0);
    if (!has) continue;
    for (size_t l = a; l < b; l++) { leaf[l].p = off[l]; leaf[l].len = len[l]; }
    it[m].leaf = leaf + a; it[m].look = look + a; it[m].n = b - a; it[m].merged = is_agg[c] && (b - a) != 1;
    pchild_prime(&it[m]);
    m++;
  }
  int r = 1;
  if (m > 1) {
    uint32_t ms = max_slop < 0 ? 0xFFFFFFFFu : (uint32_t)max_slop;
    r = in_order ? prox_in_order(it, m, ms) : prox_unordered(it, m, ms);
  }
  free(leaf); free(look); free(it);
  return r;
}
int oracle_min_offset_delta(size_t n_children, const size_t *child_first, const int *is_agg, const uint8_t *const *off,
                            const uint32_t *len) {
  if (n_children <= 1) return 1;
  size_t n_leaves = child_first[n_children];
  int dist = 0; size_t i = 0, num = n_children;
#define HAS(c) (is_agg[c] ? 1 : (child_first[(c) + 1] > child_first[c] && len[child_first[c]] > 0))
  while (i < num) {
    while (i < num && !HAS(i)) i++;
    if (i == num) break;
    size_t c1 = i++;
    while (i < num && !HAS(i)) i++;
    if (i == num) break;
    size_t c2 = i;
    PTerm *leaf = calloc(n_leaves, sizeof *leaf); uint32_t *look = calloc(n_leaves, sizeof *look);
    PChild v[2]; size_t cs[2] = {c1, c2};
    for (int k = 0; k < 2; k++) {
      size_t a = child_first[cs[k]], b = child_first[cs[k] + 1];
      for (size_t l = a; l < b; l++) { leaf[l].p = off[l]; leaf[l].len = len[l]; leaf[l].pos = 0; leaf[l].last = 0; }
      v[k].leaf = leaf + a; v[k].look = look + a; v[k].n = b - a; v[k].merged = is_agg[cs[k]] && (b - a) != 1;
      pchild_prime(&v[k]);
    }
    uint32_t p1 = pchild_next(&v[0]), p2 = pchild_next(&v[1]);
    int cd = (int)(p2 > p1 ? p2 - p1 : p1 - p2);
    while (cd > 1 && p1 != P_EOF && p2 != P_EOF) {
      uint32_t a = p2 > p1 ? p2 - p1 : p1 - p2;
      if (a < (uint32_t)cd) cd = (int)a;
      if (p2 > p1) p1 = pchild_next(&v[0]); else p2 = pchild_next(&v[1]);
    }
    dist += cd * cd;
    free(leaf); free(look);
  }
#undef HAS
  return dist ? (int)sqrt((double)dist) : (int)(num - 1);
}

/* Intersection with proximity constraints (Intersection::new_with_slop_order, reference
 * rqe_iterators/src/intersection.rs:94-119,200-245): in_order keeps the children in the caller's order (they are not
 * sorted by estimate), a consensus document that is not within range is skipped.  slops_out[hit] (optional) =
 * IndexResult_MinOffsetDelta of the hit in aggregate (iteration) order. */
size_t oracle_intersect_ex(const OInv **lists, size_t nl, size_t cap, long max_slop, int in_order, uint64_t *ids,
                           uint32_t *freqs, uint32_t *masks, int32_t *slops_out) {
  if (!nl) return 0;
  Child *c = malloc(nl * sizeof *c);
  for (size_t i = 0; i < nl; i++) { c[i].r = oreader_new(lists[i]); c[i].orig = i; c[i].cur = 0; }
  if (!in_order)
    for (size_t i = 1; i < nl; i++) {
      Child t = c[i]; size_t j = i;
      while (j && c[j - 1].r->ii->n_unique > t.r->ii->n_unique) { c[j] = c[j - 1]; j--; }
      c[j] = t;
    }
  const int check = max_slop >= 0 || in_order;
  size_t *first = malloc((nl + 1) * sizeof *first); int *agg = calloc(nl, sizeof *agg);
  const uint8_t **off = malloc(nl * sizeof *off); uint32_t *len = malloc(nl * sizeof *len);
  for (size_t i = 0; i <= nl; i++) first[i] = i;
  size_t hits = 0; int eof = 0;
  while (!eof && hits < cap) {
    if (!oreader_next(c[0].r)) break;
    uint64_t target = c[0].cur = c[0].r->doc;
    for (;;) {
      int agreed = 1;
      for (size_t i = 0; i < nl; i++) {
        if (c[i].cur == target) continue;
        if (!oreader_seek(c[i].r, target)) { eof = 1; agreed = 0; break; }
        c[i].cur = c[i].r->doc;
        if (c[i].cur != target) { target = c[i].cur; agreed = 0; break; }
      }
      if (eof || agreed) break;
    }
    if (eof) break;
    for (size_t i = 0; i < nl; i++) { off[i] = c[i].r->offs; len[i] = c[i].r->osz; }
    if (check && !oracle_within_range(nl, first, agg, off, len, max_slop, in_order)) continue;
    ids[hits] = target;
    for (size_t i = 0; i < nl; i++) {
      if (freqs) freqs[c[i].orig * cap + hits] = c[i].r->freq;
      if (masks) masks[c[i].orig * cap + hits] = c[i].r->mask;
    }
    if (slops_out) slops_out[hits] = oracle_min_offset_delta(nl, first, agg, off, len);
    hits++;
  }
  for (size_t i = 0; i < nl; i++) oreader_free(c[i].r);
  free(c); free(first); free(agg); free(off); free(len);
  return hits;
}

/* varint-delta offsets vector -> absolute token positions (RSOffsetVector iteration) */
size_t oracle_decode_offsets(const uint8_t *p, size_t len, uint32_t *out, size_t cap) {
  size_t n = 0, pos = 0; uint32_t last = 0;
  while (pos < len && n < cap) {
    uint64_t d; size_t k = oracle_varint_decode(p + pos, len - pos, &d); if (!k) break;
    last += (uint32_t)d; out[n++] = last; pos += k;
  }
  return n;
}

/* ---- union and NOT over posting lists ------------------------------------------------------------------------
 * Union (reference src/redisearch_rs/rqe_iterators/src/union_flat.rs:223-257 find the minimum doc id among
 * the children, :297-320 aggregate every child positioned on it): documents present in ANY list, ascending;
 * a list that does not hold the document contributes nothing (freq 0 / mask 0 in the [list][hit] layout).
 * Returns the number of hits (stops at cap). */
size_t oracle_union(const OInv **lists, size_t nl, size_t cap, uint64_t *ids, uint32_t *freqs, uint32_t *masks) {
  if (!nl) return 0;
  OReader **r = malloc(nl * sizeof *r);
  int *live = malloc(nl * sizeof *live);
  for (size_t i = 0; i < nl; i++) { r[i] = oreader_new(lists[i]); live[i] = oreader_next(r[i]); }
  size_t hits = 0;
  while (hits < cap) {
    uint64_t min_id = UINT64_MAX;
    for (size_t i = 0; i < nl; i++) if (live[i] && r[i]->doc < min_id) min_id = r[i]->doc;
    if (min_id == UINT64_MAX) break;
    ids[hits] = min_id;
    for (size_t i = 0; i < nl; i++) {
      int on = live[i] && r[i]->doc == min_id;
      if (freqs) freqs[i * cap + hits] = on ? r[i]->freq : 0;
      if (masks) masks[i * cap + hits] = on ? r[i]->mask : 0;
      if (on) live[i] = oreader_next(r[i]);
    }
    hits++;
  }
  for (size_t i = 0; i < nl; i++) oreader_free(r[i]);
  free(r); free(live);
  return hits;
}

/* NOT (reference src/redisearch_rs/rqe_iterators/src/not.rs:171-209: every doc id in 1..=max_doc_id the child
 * does not hold; not_optimized.rs: the same relative to a wildcard list of existing documents).  `universe`
 * may be NULL.  Results are virtual (no term data). */
size_t oracle_not(const OInv *child, const OInv *universe, uint64_t max_doc_id, size_t cap, uint64_t *ids) {
  OReader *c = oreader_new(child), *u = universe ? oreader_new(universe) : NULL;
  int clive = oreader_next(c);
  size_t hits = 0;
  if (u) {
    while (hits < cap && oreader_next(u)) {
      uint64_t d = u->doc;
      if (d > max_doc_id) break;
      while (clive && c->doc < d) clive = oreader_next(c);
      if (!(clive && c->doc == d)) ids[hits++] = d;
    }
    oreader_free(u);
  } else {
    for (uint64_t d = 1; d <= max_doc_id && hits < cap; d++) {
      while (clive && c->doc < d) clive = oreader_next(c);
      if (!(clive && c->doc == d)) ids[hits++] = d;
    }
  }
  oreader_free(c);
  return hits;
}

/* The order an intersection iterates its children in -- and the order its result holds them in, which is the order the
 * scorers add in.  Reference src/redisearch_rs/rqe_iterators/src/intersection.rs:94-119 (new_with_slop_order: a STABLE
 * sort by num_estimated() as f64 * intersection_sort_weight(), total_cmp; in_order keeps the query's order), the weights:
 * intersection.rs:580-582 (a child Intersection: 1 / children, at least one), union_flat.rs:817-823 / union_heap.rs:686-692 (a
 * child Union: its children when prioritizeIntersectUnionChildren -- off by default, src/config.h:451 -- else 1),
 * lib.rs:325-328 (everything else 1).  kind: 0 term / other, 1 union, 2 intersection. */
double oracle_intersection_sort_key(size_t num_estimated, int kind, size_t n_children, int prioritize_union_children) {
  double w = 1.0;
  size_t n = n_children > 1 ? n_children : 1;
  if (kind == 2) w = 1.0 / (double)n;
  else if (kind == 1 && prioritize_union_children) w = (double)n;
  return (double)num_estimated * w;
}
void oracle_intersection_child_order(size_t n, const size_t *num_estimated, const int *kind, const size_t *n_children,
                                     int prioritize_union_children, int in_order, size_t *order_out) {
  for (size_t i = 0; i < n; i++) order_out[i] = i;
  if (in_order) return;
  for (size_t i = 1; i < n; i++) { /* insertion sort: stable */
    size_t c = order_out[i];
    double kc = oracle_intersection_sort_key(num_estimated[c], kind[c], n_children[c], prioritize_union_children);
    size_t j = i;
    while (j > 0) {
      size_t p = order_out[j - 1];
      if (oracle_intersection_sort_key(num_estimated[p], kind[p], n_children[p], prioritize_union_children) <= kc) break;
      order_out[j] = p;
      j--;
    }
    order_out[j] = c;
  }
}
