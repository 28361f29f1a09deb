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
#include <string.h>

enum {
  C_FULL = 0, C_FREQS_FIELDS = 1, C_FREQS_ONLY = 2, C_FIELDS_ONLY = 3, C_FIELDS_OFFSETS = 4,
  C_OFFSETS_ONLY = 5, C_FREQS_OFFSETS = 6, C_DOCIDS_ONLY = 7, C_RAW_DOCIDS = 8, C_NUM = 9
};
/* qint arity and which slot holds what (-1 = absent). slot 0 is always the delta. */
typedef struct { int n, freq, mask, osz; uint16_t block_entries; } CodecDesc;
static const CodecDesc CODECS[C_NUM] = {
  {4, 1, 2, 3, 100}, {3, 1, 2, -1, 100}, {2, 1, -1, -1, 100}, {2, -1, 1, -1, 100}, {3, -1, 1, 2, 100},
  {2, -1, -1, 1, 100}, {3, 1, -1, 2, 100}, {0, -1, -1, -1, 1000}, {0, -1, -1, -1, 1000},
};

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
int oinv_add(OInv *ii, uint64_t doc, uint32_t freq, uint32_t mask, const uint8_t *offs, uint32_t osz) {
  const CodecDesc *cd = &CODECS[ii->codec];
  if (ii->nb && ii->b[ii->nb - 1].last == doc && ii->b[ii->nb - 1].n) return 0;
  OBlock *bl = (ii->nb && ii->b[ii->nb - 1].n < cd->block_entries) ? &ii->b[ii->nb - 1] : new_block(ii, doc);
  uint64_t base = (ii->codec == C_RAW_DOCIDS) ? bl->first : bl->last;
  uint64_t delta = doc - base;
  if (delta > 0xFFFFFFFFull) { bl = new_block(ii, doc); delta = 0; }
  size_t need = bl->len + 32 + (cd->osz >= 0 ? osz : 0);
  if (need > bl->cap) { bl->cap = need * 2; bl->buf = realloc(bl->buf, bl->cap); }
  uint8_t *w = bl->buf + bl->len;
  if (ii->codec == C_DOCIDS_ONLY) bl->len += oracle_varint_encode(w, delta);
  else if (ii->codec == C_RAW_DOCIDS) { uint32_t d = (uint32_t)delta; memcpy(w, &d, 4); bl->len += 4; }
  else {
    uint32_t v[4]; v[0] = (uint32_t)delta;
    if (cd->freq >= 0) v[cd->freq] = freq;
    if (cd->mask >= 0) v[cd->mask] = mask;
    if (cd->osz >= 0) v[cd->osz] = osz;
    size_t k = oracle_qint_encode(w, v, cd->n);
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
  r->freq = 0; r->mask = 0; r->offs = NULL; r->osz = 0;
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
  if (k + osz > avail) return 0;
  r->doc = base + v[0];
  if (cd->freq >= 0) r->freq = v[cd->freq];
  if (cd->mask >= 0) r->mask = v[cd->mask];
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
