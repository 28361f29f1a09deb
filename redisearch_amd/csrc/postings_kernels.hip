// postings_kernels.hip -- the integer half of the hot path on gfx950: posting-list decode, N-way
// intersection and the built-in scorers (hand-written HIP, integer / fp64 work, HBM- and
// latency-bound -- no MFMA anywhere near it).
//
// Decode follows the reference's record layouts byte for byte:
//   qint   reference src/redisearch_rs/qint/src/lib.rs:139-214 (header byte, 2 bits per value = len-1,
//          little-endian payloads)
//   varint reference src/redisearch_rs/varint/src/lib.rs (7-bit groups, MSB first, +1 per continuation)
//   blocks reference src/redisearch_rs/inverted_index/src/index/core.rs:76-96 and reader/core.rs
//          (delta base = previous doc id, the first record of a block is relative to first_doc_id;
//          RawDocIdsOnly deltas are all relative to first_doc_id)
// Intersection is set-equivalent to Intersection::find_consensus
// (reference src/redisearch_rs/rqe_iterators/src/intersection.rs:256-288): the shortest list drives
// and every other list is searched for the candidate -- here all candidates at once, a lower-bound
// binary search per (candidate, list) instead of skip_to's block search + sequential decode.
// Scorers mirror reference src/ext/default.c:68-461 with its exact float/double promotions; this
// file is compiled with -ffp-contract=off so that no multiply-add is fused behind the C source's back.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "postings_ops.hpp"
#include "search_kernels.hpp"

namespace rsgpu {

CodecDesc codec_desc(int codec) {
  static const CodecDesc T[13] = {
      {0, 4, 1, 2, 3, 0},  {0, 3, 1, 2, -1, 0}, {0, 2, 1, -1, -1, 0}, {0, 2, -1, 1, -1, 0}, {0, 3, -1, 1, 2, 0},
      {0, 2, -1, -1, 1, 0}, {0, 3, 1, -1, 2, 0}, {1, 0, -1, -1, -1, 0}, {2, 0, -1, -1, -1, 0},
      // wide: FullWide qint[delta,freq,offsetsLen] + varint mask + offsets; FreqsFieldsWide qint[delta,freq] + varint mask;
      // FieldsOnlyWide varint delta + varint mask; FieldsOffsetsWide qint[delta,offsetsLen] + varint mask + offsets
      {0, 3, 1, -1, 2, 1}, {0, 2, 1, -1, -1, 1}, {1, 0, -1, -1, -1, 1}, {0, 2, -1, -1, 1, 1}};
  if (codec < 0 || codec > 12) return CodecDesc{-1, 0, -1, -1, -1, 0};
  return T[codec];
}

namespace {

// ---- decode ----------------------------------------------------------------------------------------
// Records are variable-length and delta-coded, so a block is decoded sequentially by ONE lane -- but the 64
// blocks of a wavefront are consecutive in the byte stream, so the wavefront first copies its whole byte
// range into LDS with coalesced 16-byte loads and every lane then parses its block out of LDS (a lane
// reading its bytes straight from global memory costs one cache-line lookup per byte and lane:
// 94 GB/s of encoded bytes, profiles/r01_hybrid_config5_stages.json).  A wavefront whose byte range does
// not fit (huge offset vectors) parses from global memory as before.
// One kernel per record kind (0 qint, 1 varint delta, 2 raw u32 delta): keeping the three decoders in
// one body made hipcc (ROCm 7.2) drop the cursor advance of the raw path.
constexpr uint32_t kDecodeLds = 30 * 1024;  // bytes of encoded input staged per wavefront (5 wavefronts per CU)
constexpr uint32_t kSyncSeg = 16, kSyncPts = 7;  // sync points: before records 16, 32, .. 112 of a block (kSyncPts + 1 = 8 lanes)

// 7-bit groups, most significant first, +1 per continuation (reference varint/src/lib.rs); 128-bit accumulator
template <typename Bytes>
__device__ __forceinline__ uint32_t read_varint128(Bytes bytes, uint32_t pos, uint64_t &lo, uint64_t &hi) {
  uint32_t c = bytes[pos++];
  lo = c & 0x7fu;
  hi = 0;
  while (c & 0x80u) {
    lo++;
    if (lo == 0) hi++;
    c = bytes[pos++];
    hi = (hi << 7) | (lo >> 57);
    lo = (lo << 7) | (uint64_t)(c & 0x7fu);
  }
  return pos;
}

// Four bytes at an arbitrary position.  From LDS: two aligned dword reads + v_alignbyte_b32 -- ONE LDS round trip per
// value instead of one per byte (a lane's byte reads are dependent ds_read_u8's: ~9 round trips per qint record was what
// bounded this kernel, 140 GB/s of encoded bytes).  From global memory: byte loads as before (rare fallback).
struct LdsBytes {
  const uint8_t *p;  // 4-byte aligned, 8 readable bytes behind the last record
  __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return p[i]; }
};
__device__ __forceinline__ uint32_t peek32(LdsBytes b, uint32_t pos) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(b.p) + (pos >> 2);
  return __builtin_amdgcn_alignbyte(w[1], w[0], pos & 3u);
}
__device__ __forceinline__ uint32_t peek32(const uint8_t *b, uint32_t pos) {
  return (uint32_t)b[pos] | ((uint32_t)b[pos + 1] << 8) | ((uint32_t)b[pos + 2] << 16) | ((uint32_t)b[pos + 3] << 24);
}

template <int KIND, typename Bytes>
__device__ __forceinline__ void decode_one_block(const CodecDesc &cd, Bytes bytes, uint32_t pos, uint32_t fin,
                                                 uint32_t n, uint32_t f0, uint32_t out, uint32_t *__restrict__ ids,
                                                 uint32_t *__restrict__ freqs, uint32_t *__restrict__ masks,
                                                 uint32_t *__restrict__ wmasks, uint32_t *__restrict__ off_pos,
                                                 uint32_t *__restrict__ off_len, uint32_t abs_base,
                                                 uint32_t *__restrict__ sync_w = nullptr, uint32_t pos0 = 0) {
  uint32_t base = f0;
  // Decoded fields are collected four records at a time and written with one 16-byte store per array: a lane's entries
  // are consecutive in the output arrays, those of its neighbours ~100 entries away, so a scalar store per record is 64
  // separate 4-byte write transactions per instruction -- it was the stores, not the parsing, that bounded this kernel.
  typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
  uint32_t b_id[4], b_fr[4], b_mk[4], b_op[4], b_ol[4];
  for (uint32_t e = 0; e < n && pos < fin; e++, out++) {
    // (the sync points of a block this loop parses -- a wavefront whose 64 blocks did not fit the staging buffer: without
    // them the eight lanes that share the block next time start from whatever the allocation held)
    if (sync_w && e && (e & (kSyncSeg - 1)) == 0 && e / kSyncSeg <= kSyncPts) {
      sync_w[2 * (e / kSyncSeg - 1)] = pos - pos0;
      sync_w[2 * (e / kSyncSeg - 1) + 1] = base;
    }
    uint32_t freq = 0, mask = 0, osz = 0;
    uint64_t mlo = 0, mhi = 0;
    if (KIND == 0) {
      // the control byte fixes where every field starts, so the fields are fetched independently of each other:
      // two dependent round trips per record (control byte, then all fields)
      const uint32_t head = peek32(bytes, pos);
      const uint32_t hdr = head & 0xffu;
      pos++;
      uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (i < cd.n) {
          const uint32_t len = ((hdr >> (2 * i)) & 3u) + 1u;
          // field 0 of a record that is at most 4 bytes long came with the control byte
          const uint32_t raw = i == 0 && len < 4 ? head >> 8 : peek32(bytes, pos);
          v[i] = len == 4 ? raw : raw & ((1u << (8 * len)) - 1u);
          pos += len;
        }
      }
      base += v[0];
      if (cd.freq >= 0) freq = cd.freq == 1 ? v[1] : (cd.freq == 2 ? v[2] : v[3]);
      if (cd.mask >= 0) mask = cd.mask == 1 ? v[1] : (cd.mask == 2 ? v[2] : v[3]);
      if (cd.osz >= 0) osz = cd.osz == 1 ? v[1] : (cd.osz == 2 ? v[2] : v[3]);
    } else if (KIND == 1) {
      // a 32-bit delta takes at most five bytes: parsed out of two fetched words, no memory access inside the loop
      uint64_t win = (uint64_t)peek32(bytes, pos) | ((uint64_t)peek32(bytes, pos + 4) << 32);
      uint32_t c = (uint32_t)win & 0xffu;
      uint32_t val = c & 0x7fu;
      pos++;
      while (c & 0x80u) {
        val++;
        win >>= 8;
        c = (uint32_t)win & 0xffu;
        pos++;
        val = (val << 7) | (c & 0x7fu);
      }
      base += val;
    } else {
      const uint32_t d = peek32(bytes, pos);
      pos += 4;
      base = f0 + d;
    }
    if (cd.wide) {
      pos = read_varint128(bytes, pos, mlo, mhi);
      mask = (uint32_t)mlo;
    }
    const uint32_t slot = e & 3u;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++)
      if (slot == j) {
        b_id[j] = base;
        b_fr[j] = freq;
        b_mk[j] = mask;
        b_op[j] = abs_base + pos;
        b_ol[j] = osz;
      }
    const bool last = e + 1 == n || pos + osz >= fin;
    if (slot == 3u) {  // records e-3 .. e -> entries out-3 .. out
      const uint32_t o = out - 3u;
      *reinterpret_cast<u4u *>(ids + o) = (u4u){b_id[0], b_id[1], b_id[2], b_id[3]};
      if (freqs) *reinterpret_cast<u4u *>(freqs + o) = (u4u){b_fr[0], b_fr[1], b_fr[2], b_fr[3]};
      if (masks) *reinterpret_cast<u4u *>(masks + o) = (u4u){b_mk[0], b_mk[1], b_mk[2], b_mk[3]};
      if (off_pos) {  // where the offsets blob of each record sits in the list's byte buffer
        *reinterpret_cast<u4u *>(off_pos + o) = (u4u){b_op[0], b_op[1], b_op[2], b_op[3]};
        *reinterpret_cast<u4u *>(off_len + o) = (u4u){b_ol[0], b_ol[1], b_ol[2], b_ol[3]};
      }
    } else if (last) {  // the block's last one to three records
#pragma unroll
      for (uint32_t j = 0; j < 3; j++)
        if (j <= slot) {
          const uint32_t o = out - slot + j;
          ids[o] = b_id[j];
          if (freqs) freqs[o] = b_fr[j];
          if (masks) masks[o] = b_mk[j];
          if (off_pos) {
            off_pos[o] = b_op[j];
            off_len[o] = b_ol[j];
          }
        }
    }
    if (wmasks) {
      wmasks[4 * (size_t)out] = (uint32_t)mlo;
      wmasks[4 * (size_t)out + 1] = (uint32_t)(mlo >> 32);
      wmasks[4 * (size_t)out + 2] = (uint32_t)mhi;
      wmasks[4 * (size_t)out + 3] = (uint32_t)(mhi >> 32);
    }
    pos += osz;  // offsets bytes are not parsed here: the proximity kernels read them in place
  }
}

// The qint codecs without a wide mask, parsed out of LDS with the record layout a compile-time constant (NF fields; FR / MK /
// OS = which of fields 1..3 is the frequency / field mask / offsets length, -1 none): ONE LDS round trip per record -- the
// NF + 2 aligned words that hold it whatever its length -- and straight-line field extraction.  The generic loop above
// branches on the codec description per field and waits for up to three dependent LDS reads per record; a lane's 100
// records are a serial chain, the lists of a query leave most SIMDs with a single wavefront, so the kernel's time IS that
// chain: 52-83 us for a 2.5 M / 5 M-entry list with the generic loop, 20-26 us with this one (profiles/r03_decode.txt; a
// register-FIFO variant that keeps the LDS reads out of the chain altogether was tried and was slower: 64-bit funnel
// shifts cost more issue slots than the round trips they save).  Same bytes in, same arrays out
// (reference qint/src/lib.rs:139-214; inverted_index/src/codec/{freqs_only,freqs_fields,full,...}.rs).
template <int NF, int FR, int MK, int OS>
__device__ __forceinline__ void decode_qint_block_lds(const uint8_t *stage, uint32_t pos, uint32_t fin, uint32_t n,
                                                      uint32_t f0, uint32_t out, uint32_t *__restrict__ ids,
                                                      uint32_t *__restrict__ freqs, uint32_t *__restrict__ masks,
                                                      uint32_t *__restrict__ off_pos, uint32_t *__restrict__ off_len,
                                                      uint32_t abs_base, uint32_t *__restrict__ sync_w, uint32_t pos0) {
  typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
  uint32_t base = f0, e = 0;
  while (e < n && pos < fin) {
    if (sync_w && e && (e & (kSyncSeg - 1)) == 0 && e / kSyncSeg <= kSyncPts) {  // record e starts here, after doc id `base`
      sync_w[2 * (e / kSyncSeg - 1)] = pos - pos0;
      sync_w[2 * (e / kSyncSeg - 1) + 1] = base;
    }
    uint32_t b_id[4], b_fr[4], b_mk[4], b_op[4], b_ol[4];
    uint32_t got = 0;
    if constexpr (OS < 0) {
      // FOUR records at once when their control bytes are all zero (round 6): every field is one byte then, the records are
      // NF + 1 bytes each and sit at fixed offsets in the NF + 1 words one LDS round trip delivers anyway -- no position depends
      // on a value, ~8 vector instructions per record instead of ~37 and a quarter of the round trips.  That is the usual
      // record of the lists that cost the most to decode: dense ones (deltas and frequencies below 256).  Any other control
      // byte among the four, a block end or fewer than four records left: the record-by-record path below, same bytes out.
      constexpr uint32_t RL = NF + 1;
      if (e + 4 <= n && pos + 4 * RL <= fin) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(stage) + (pos >> 2);
        uint32_t r[NF + 2];
#pragma unroll
        for (int i = 0; i < NF + 2; i++) r[i] = w[i];
        uint32_t a[NF + 1];
#pragma unroll
        for (int i = 0; i < NF + 1; i++) a[i] = __builtin_amdgcn_alignbyte(r[i + 1], r[i], pos & 3u);
        uint32_t ctrl = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) ctrl |= a[(i * RL) >> 2] & (0xFFu << (8 * ((i * RL) & 3u)));
        if (ctrl == 0) {
          auto byte_at = [&](uint32_t b) { return (a[b >> 2] >> (8 * (b & 3u))) & 0xFFu; };
#pragma unroll
          for (uint32_t i = 0; i < 4; i++) {
            base += byte_at(i * RL + 1);
            b_id[i] = base;
            b_fr[i] = FR >= 0 ? byte_at(i * RL + 1 + (FR >= 0 ? FR : 0)) : 0u;
            b_mk[i] = MK >= 0 ? byte_at(i * RL + 1 + (MK >= 0 ? MK : 0)) : 0u;
            b_op[i] = abs_base + pos + (i + 1) * RL;
            b_ol[i] = 0u;
          }
          pos += 4 * RL;
          e += 4;
          *reinterpret_cast<u4u *>(ids + out) = (u4u){b_id[0], b_id[1], b_id[2], b_id[3]};
          if (freqs) *reinterpret_cast<u4u *>(freqs + out) = (u4u){b_fr[0], b_fr[1], b_fr[2], b_fr[3]};
          if (masks) *reinterpret_cast<u4u *>(masks + out) = (u4u){b_mk[0], b_mk[1], b_mk[2], b_mk[3]};
          if (off_pos) {
            *reinterpret_cast<u4u *>(off_pos + out) = (u4u){b_op[0], b_op[1], b_op[2], b_op[3]};
            *reinterpret_cast<u4u *>(off_len + out) = (u4u){b_ol[0], b_ol[1], b_ol[2], b_ol[3]};
          }
          out += 4;
          continue;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (e < n && pos < fin) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(stage) + (pos >> 2);
        uint32_t r[NF + 2];
#pragma unroll
        for (int i = 0; i < NF + 2; i++) r[i] = w[i];
        const uint32_t sh = pos & 3u;
        uint32_t a[NF + 1];  // the record from its control byte on
#pragma unroll
        for (int i = 0; i < NF + 1; i++) a[i] = __builtin_amdgcn_alignbyte(r[i + 1], r[i], sh);
        const uint32_t hdr = a[0];
        uint32_t v[NF], o = 1;
        if ((hdr & 0xFFu) == 0 && OS >= 0) {
          // a zero control byte: every field is the byte at its fixed place (round 6; the records of a dense list with inline
          // offsets -- their position still depends on the offsets lengths before them, so one at a time, but a dozen dependent
          // instructions instead of fifty; the layouts WITHOUT offsets take four such records at once above)
#pragma unroll
          for (int i = 0; i < NF; i++) v[i] = (a[(1 + i) >> 2] >> (8 * ((1 + i) & 3))) & 0xFFu;
          o = 1 + NF;
        } else {
#pragma unroll
          for (int i = 0; i < NF; i++) {
            const uint32_t len = ((hdr >> (2 * i)) & 3u) + 1u;
            // four bytes from byte o of the record: o <= 1 + 4 i, so they start in word 0 .. i
            uint32_t raw = __builtin_amdgcn_alignbyte(a[1], a[0], o);
#pragma unroll
            for (int d = 1; d <= i; d++) raw = (o >> 2) == (uint32_t)d ? __builtin_amdgcn_alignbyte(a[d + 1], a[d], o) : raw;
            const uint32_t drop = 32u - 8u * len;  // keep the low `len` bytes (a shift by 0 when len == 4)
            v[i] = (raw << drop) >> drop;
            o += len;
          }
        }
        pos += o;
        base += v[0];
        b_id[j] = base;
        b_fr[j] = FR >= 0 ? v[FR >= 0 ? FR : 0] : 0u;
        b_mk[j] = MK >= 0 ? v[MK >= 0 ? MK : 0] : 0u;
        b_op[j] = abs_base + pos;
        b_ol[j] = OS >= 0 ? v[OS >= 0 ? OS : 0] : 0u;
        if (OS >= 0) pos += b_ol[j];  // offsets bytes are not parsed here: the proximity kernels read them in place
        e++;
        got = j + 1;
      }
    }
    if (got == 4) {
      *reinterpret_cast<u4u *>(ids + out) = (u4u){b_id[0], b_id[1], b_id[2], b_id[3]};
      if (freqs) *reinterpret_cast<u4u *>(freqs + out) = (u4u){b_fr[0], b_fr[1], b_fr[2], b_fr[3]};
      if (masks) *reinterpret_cast<u4u *>(masks + out) = (u4u){b_mk[0], b_mk[1], b_mk[2], b_mk[3]};
      if (off_pos) {
        *reinterpret_cast<u4u *>(off_pos + out) = (u4u){b_op[0], b_op[1], b_op[2], b_op[3]};
        *reinterpret_cast<u4u *>(off_len + out) = (u4u){b_ol[0], b_ol[1], b_ol[2], b_ol[3]};
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < 3; j++)
        if (j < got) {
          ids[out + j] = b_id[j];
          if (freqs) freqs[out + j] = b_fr[j];
          if (masks) masks[out + j] = b_mk[j];
          if (off_pos) {
            off_pos[out + j] = b_op[j];
            off_len[out + j] = b_ol[j];
          }
        }
    }
    out += got;
  }
}

// cd -> the instantiation above (false: not one of the reference's qint layouts, the generic loop takes it)
__device__ __forceinline__ bool decode_qint_fast(const CodecDesc &cd, const uint8_t *stage, uint32_t pos, uint32_t fin, uint32_t n,
                                                 uint32_t f0, uint32_t out, uint32_t *ids, uint32_t *freqs, uint32_t *masks,
                                                 uint32_t *off_pos, uint32_t *off_len, uint32_t abs_base, uint32_t *sync_w,
                                                 uint32_t pos0) {
#define RSGPU_QINT(NF, FR, MK, OS)                                                                                         \
  if (cd.n == NF && cd.freq == FR && cd.mask == MK && cd.osz == OS) {                                                      \
    decode_qint_block_lds<NF, FR, MK, OS>(stage, pos, fin, n, f0, out, ids, freqs, masks, off_pos, off_len, abs_base,      \
                                          sync_w, pos0);                                                                   \
    return true;                                                                                                           \
  }
  RSGPU_QINT(2, 1, -1, -1)  // FreqsOnly
  RSGPU_QINT(2, -1, 1, -1)  // FieldsOnly
  RSGPU_QINT(2, -1, -1, 1)  // OffsetsOnly
  RSGPU_QINT(3, 1, 2, -1)   // FreqsFields
  RSGPU_QINT(3, 1, -1, 2)   // FreqsOffsets
  RSGPU_QINT(3, -1, 1, 2)   // FieldsOffsets
  RSGPU_QINT(4, 1, 2, 3)    // Full
#undef RSGPU_QINT
  return false;
}

// sync (qint layouts without a wide mask): [n_blocks][kSyncPts][2] = {byte offset from the block's first byte, doc id before
// it} of records 16, 32, ... of every block.  sync_mode 1: the lane that parses a block writes them as it passes; sync_mode
// 2: they are there, and a block is parsed by EIGHT lanes, 16 records each -- a lane's 100 records were the one serial chain
// this kernel's time consists of (parse ~18 us of 25, whatever the list length: profiles/r03_decode.txt), a list is immutable
// after upload, so the first decode of a list leaves the sync points behind for all later ones (8 bytes per 16 postings).
struct DecodeArgs {  // one list's decode_blocks_kernel arguments (two of them: decode_blocks_pair_kernel)
  CodecDesc cd;
  const uint8_t *bytes;
  const uint64_t *byte_off;
  const uint32_t *first, *nent, *entry_off;
  uint32_t n_blocks;
  uint32_t *ids, *freqs, *masks, *wmasks, *off_pos, *off_len, *sync;
  int sync_mode;
  uint32_t lds_cap;
  uint32_t bpw1;  // blocks per wavefront when a lane parses a whole block (sync_mode 0 / 1): 64, or fewer for lists whose blocks
                  // are long (inline offsets) so that a wavefront's byte range still fits the staging buffer
};

template <int KIND>
__device__ __forceinline__ void decode_blocks_body(const DecodeArgs &A, uint32_t wg, uint8_t *stage) {
  const CodecDesc cd = A.cd;
  const uint8_t *__restrict__ bytes = A.bytes;
  const uint64_t *__restrict__ byte_off = A.byte_off;
  const uint32_t *__restrict__ first = A.first;
  const uint32_t *__restrict__ nent = A.nent;
  const uint32_t *__restrict__ entry_off = A.entry_off;
  const uint32_t n_blocks = A.n_blocks;
  uint32_t *__restrict__ ids = A.ids;
  uint32_t *__restrict__ freqs = A.freqs;
  uint32_t *__restrict__ masks = A.masks;
  uint32_t *__restrict__ wmasks = A.wmasks;
  uint32_t *__restrict__ off_pos = A.off_pos;
  uint32_t *__restrict__ off_len = A.off_len;
  uint32_t *__restrict__ sync = A.sync;
  const int sync_mode = A.sync_mode;
  const uint32_t lds_cap = A.lds_cap;
  const uint32_t lane = threadIdx.x;
  const uint32_t lpb = sync_mode == 2 ? kSyncPts + 1 : 1, bpw = sync_mode == 2 ? 64 / lpb : A.bpw1;  // lanes per block, blocks per wavefront
  const uint32_t b0 = wg * bpw;
  const uint32_t nb = n_blocks - b0 < bpw ? n_blocks - b0 : bpw;
  // every lane's block description in ONE memory round trip, before anything depends on it (the wavefront's byte range
  // is the first lane's start .. the last lane's end: no separate loads for it)
  const uint32_t part = sync_mode == 2 ? lane & kSyncPts : 0, lb = sync_mode == 2 ? lane / (kSyncPts + 1) : lane;
  const uint32_t b = b0 + (lb < nb ? lb : nb - 1);
  const uint64_t beg = byte_off[b], fin = byte_off[b + 1];
  uint32_t my_n = nent[b], my_first = first[b], my_out = entry_off[b], my_skip = 0;
  if (part) {
    my_skip = sync[((size_t)b * kSyncPts + part - 1) * 2];
    my_first = sync[((size_t)b * kSyncPts + part - 1) * 2 + 1];
  }
  if (sync_mode == 2) {  // records [16 part, 16 part + 16) of the block; the last lane takes whatever is left
    const uint32_t done = part * kSyncSeg;
    my_out += done;
    my_n = my_n > done ? (my_n - done < kSyncSeg || part == kSyncPts ? my_n - done : kSyncSeg) : 0;
  }
  const uint64_t w_beg = __shfl(beg, 0) & ~15ull, w_end = __shfl(fin, (int)(nb * lpb) - 1);  // 16-byte aligned start
  const bool staged = w_end - w_beg <= lds_cap;                                               // wave-uniform
  if (staged) {
    // global -> LDS by DMA, 1 KiB per instruction, every piece in flight before the one wait: a load / ds_write loop is
    // serialised by hipcc (each store waits for its load), which made ~20 dependent HBM round trips the longest thing
    // this kernel did.  (The byte buffer carries 16 bytes of slack; lanes past the span stay off.)
    const uint32_t span = (uint32_t)(w_end - w_beg);
    const uint8_t *src = bytes + w_beg + lane * 16;
    for (uint32_t o = 0; o < span; o += 64 * 16) {
      if (o + lane * 16 < span)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + o),
                                         (__attribute__((address_space(3))) void *)(stage + o), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (sync_mode == 2 && !staged) {  // (blocks too large to stage: whole blocks through the generic loop, one lane each)
    if (part) return;
    my_n = nent[b];
  }
  if (lb >= nb || !my_n) return;
  uint32_t *sync_w = sync_mode == 1 ? sync + (size_t)b * kSyncPts * 2 : nullptr;
  if (staged) {
    const uint32_t pos0 = (uint32_t)(beg - w_beg);
    if (KIND == 0 && !cd.wide && !wmasks &&
        decode_qint_fast(cd, stage, pos0 + my_skip, (uint32_t)(fin - w_beg), my_n, my_first, my_out, ids, freqs, masks, off_pos,
                         off_len, (uint32_t)w_beg, sync_w, pos0))
      return;
    decode_one_block<KIND>(cd, LdsBytes{stage}, pos0, (uint32_t)(fin - w_beg), my_n, my_first, my_out, ids, freqs, masks, wmasks,
                           off_pos, off_len, (uint32_t)w_beg, sync_w, pos0);
  }
  else  // positions relative to the block start stay below 2^32 (a block holds <= 1000 records)
    decode_one_block<KIND>(cd, bytes + beg, 0u, (uint32_t)(fin - beg), my_n, my_first, my_out, ids, freqs, masks, wmasks, off_pos,
                           off_len, (uint32_t)beg, sync_w, 0u);
}

template <int KIND>
__global__ __launch_bounds__(64) void decode_blocks_kernel(DecodeArgs a) {
  // lds_cap bytes of staging + 64 of slack (the parsers fetch whole words ahead); dynamic: with eight lanes per block a
  // wavefront stages a few KiB, and 30 KiB each would leave five wavefronts per CU
  extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
  decode_blocks_body<KIND>(a, blockIdx.x, stage);
}

// Two lists, one launch (the lists of a query decoded per query -- cache_decoded = 0 -- cost ~7 us of fixed time per launch
// next to 13-20 us of work each, profiles/r03_decode.txt): workgroups [0, wgs_a) take list a, the rest list b.
template <int KIND>
__global__ __launch_bounds__(64) void decode_blocks_pair_kernel(DecodeArgs a, DecodeArgs b, uint32_t wgs_a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
  if (blockIdx.x < wgs_a) decode_blocks_body<KIND>(a, blockIdx.x, stage);
  else decode_blocks_body<KIND>(b, blockIdx.x - wgs_a, stage);
}

// ---- round 6: DENSE groups -- sixteen consecutive blocks whose records are all of the minimal length ----
// A qint record is one control byte + 1..4 bytes per field; a block whose byte length is (NF + 1) x its entry count can only hold
// records of NF + 1 bytes -- every control byte zero, every field one byte -- and that is what the lists that cost the most to
// decode look like (a term in a tenth of the documents: deltas ~10, frequencies 1..3: configs[4]'s lists are 3.00 bytes per posting).
// Then NOTHING depends on a parse: record i of the group sits at byte (NF + 1) i.  A wavefront takes sixteen blocks (lane j < 16 holds
// block j's description -- no sync points read), stages their bytes by DMA, and in steps of 256 records every lane decodes FOUR
// consecutive records: one LDS round trip, byte extracts, a wave-wide prefix of the deltas; the doc ids are the prefix + a per-block
// constant (the block's first id - the prefix in front of it: wave-uniform, taken with v_readlane from the lane that holds the
// block's first record).  Adjacent lanes write adjacent 16 bytes: a store instruction covers 1 KiB of consecutive addresses where
// the chain parsers write 64 pieces 64 bytes apart (64 partial-line requests per instruction -- the L2's request rate, not the
// parse, was what the two-round kernel's time had come down to once four zero-control records were taken at a time).
// A group with any longer record, or one too wide to stage: the eight-lanes-per-block form, its two halves one after the other.
// Same bytes in, same arrays out (reference qint/src/lib.rs:139-214; inverted_index/src/codec/{freqs_only,fields_only,freqs_fields}.rs).
constexpr uint32_t kDenseBlocks = 16;  // (32 measured: 26-27 us where 16 take 22-23 -- a wavefront's steps are one after the other; 8 would be two rounds over the chip again)
constexpr uint32_t kDenseSpans = kDenseBlocks / (64u / (kSyncPts + 1));  // eight-block spans (what lds_cap holds) per group
template <int NF, int FR, int MK>
__device__ __forceinline__ bool decode_dense_group(const DecodeArgs &A, uint32_t g16, uint8_t *stage, uint32_t stage_cap) {
  typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr uint32_t RL = NF + 1;
  const uint32_t lane = threadIdx.x;
  const uint32_t b0 = g16 * kDenseBlocks;
  const uint32_t nb = A.n_blocks - b0 < kDenseBlocks ? A.n_blocks - b0 : kDenseBlocks;
  const uint32_t b = b0 + (lane < nb ? lane : nb - 1);
  const uint64_t beg = A.byte_off[b], fin = A.byte_off[b + 1];
  const uint32_t n = A.nent[b], first = A.first[b], out = A.entry_off[b];
  // every record minimal, the blocks' entries consecutive in the arrays (they are: entry_off is a running sum -- checked, not assumed)
  const uint32_t out_next = __shfl_down(out, 1, 64);
  const bool bad = fin - beg != (uint64_t)RL * n || (lane + 1 < nb && out + n != out_next);
  if (__ballot(bad)) return false;
  const uint64_t beg0 = __shfl(beg, 0, 64), w_beg = beg0 & ~15ull, w_end = __shfl(fin, (int)nb - 1, 64);
  if (w_end - w_beg > stage_cap) return false;
  const uint32_t span = (uint32_t)(w_end - w_beg);
  {
    const uint8_t *src = A.bytes + w_beg + lane * 16;
    for (uint32_t o = 0; o < span; o += 64 * 16)
      if (o + lane * 16 < span)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + o),
                                         (__attribute__((address_space(3))) void *)(stage + o), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const uint32_t out0 = __shfl(out, 0, 64);
  const uint32_t R = __shfl(out, (int)nb - 1, 64) + __shfl(n, (int)nb - 1, 64) - out0;  // records of the group
  const uint32_t byte0 = (uint32_t)(beg0 - w_beg);
  const uint32_t start = out - out0;  // lane j < nb: the group-relative index of block j's first record
  uint32_t carry = 0;                 // the deltas of every record in front of this step (wave-uniform)
  uint32_t cur_c = 0;                 // first id - prefix in front, of the block the step's first record is in
  uint32_t jn = 0;                    // the next block whose first record has not been passed
  uint32_t s_next = __builtin_amdgcn_readlane((int)start, 0);
  for (uint32_t t0 = 0; t0 < R; t0 += 256) {
    const uint32_t i0 = t0 + 4 * lane;
    const uint32_t boff = byte0 + RL * (i0 < R ? i0 : 0u);  // (a lane past the end reads the group's first record: values unused)
    const uint32_t *w = reinterpret_cast<const uint32_t *>(stage) + (boff >> 2);
    uint32_t r[NF + 2];
#pragma unroll
    for (int i = 0; i < NF + 2; i++) r[i] = w[i];  // (past the group's last byte: inside the staging buffer's slack, unused)
    uint32_t a[NF + 1];
#pragma unroll
    for (int i = 0; i < NF + 1; i++) a[i] = __builtin_amdgcn_alignbyte(r[i + 1], r[i], boff & 3u);
    auto byte_at = [&](uint32_t x) { return (a[x >> 2] >> (8 * (x & 3u))) & 0xFFu; };
    uint32_t p[4], fr[4], mk[4];
    uint32_t run = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      run += i0 + k < R ? byte_at(k * RL + 1) : 0u;
      p[k] = run;  // inclusive, within the lane
      fr[k] = FR >= 0 ? byte_at(k * RL + 1 + (FR >= 0 ? FR : 0)) : 0u;
      mk[k] = MK >= 0 ? byte_at(k * RL + 1 + (MK >= 0 ? MK : 0)) : 0u;
    }
    // inclusive over the lanes: four shifts inside a row of sixteen, then the rows' last lanes broadcast to the rows behind them --
    // six DPP adds (a lane without a source adds 0); __shfl_up would be six dependent ds_bpermute round trips
    // inclusive over the lanes: four shifts inside a row of sixteen, then the rows' last lanes broadcast to the rows behind them --
    // six DPP adds (a lane without a source adds 0) instead of six ds_bpermute round trips (measured the same: the kernel's time is
    // its three dependent round trips -- descriptions, staging, stores -- and the launch of 4 688 workgroups, not this)
    uint32_t inc = run;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);  // row_shr:1
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);  // row_shr:2
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);  // row_shr:4
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);  // row_shr:8
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    const uint32_t exc = inc - run + carry;  // everything in front of the lane's first record
    // the blocks that begin inside this step: their constants, and which of the lane's records they own
    uint32_t base[4] = {cur_c, cur_c, cur_c, cur_c};
    while (jn < nb && s_next < t0 + 256) {  // (wave-uniform)
      const uint32_t rel = s_next - t0, lj = rel >> 2, kj = rel & 3u;
      const uint32_t e_lane = (uint32_t)__builtin_amdgcn_readlane((int)exc, lj);
      const uint32_t e_in = kj == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)(kj == 1 ? p[0] : (kj == 2 ? p[1] : p[2])), lj);
      const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)first, jn) - (e_lane + e_in);
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) base[k] = 4 * lane + k >= rel ? c : base[k];
      cur_c = c;
      jn++;
      s_next = jn < nb ? (uint32_t)__builtin_amdgcn_readlane((int)start, jn) : 0xFFFFFFFFu;
    }
    carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    uint32_t id[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) id[k] = exc + p[k] + base[k];
    const uint32_t o = out0 + i0;
    if (i0 + 4 <= R) {
      *reinterpret_cast<u4u *>(A.ids + o) = (u4u){id[0], id[1], id[2], id[3]};
      if (A.freqs) *reinterpret_cast<u4u *>(A.freqs + o) = (u4u){fr[0], fr[1], fr[2], fr[3]};
      if (A.masks) *reinterpret_cast<u4u *>(A.masks + o) = (u4u){mk[0], mk[1], mk[2], mk[3]};
    } else {
#pragma unroll
      for (uint32_t k = 0; k < 3; k++)
        if (i0 + k < R) {
          A.ids[o + k] = id[k];
          if (A.freqs) A.freqs[o + k] = fr[k];
          if (A.masks) A.masks[o + k] = mk[k];
        }
    }
  }
  return true;
}
__device__ __forceinline__ void decode_dense_body(const DecodeArgs &A, uint32_t g16, uint8_t *stage, uint32_t stage_cap) {
  const CodecDesc cd = A.cd;
  bool done = false;
  if (cd.n == 2 && cd.freq == 1 && cd.mask == -1) done = decode_dense_group<2, 1, -1>(A, g16, stage, stage_cap);        // FreqsOnly
  else if (cd.n == 2 && cd.freq == -1 && cd.mask == 1) done = decode_dense_group<2, -1, 1>(A, g16, stage, stage_cap);   // FieldsOnly
  else if (cd.n == 3 && cd.freq == 1 && cd.mask == 2) done = decode_dense_group<3, 1, 2>(A, g16, stage, stage_cap);    // FreqsFields
  if (done) return;
  constexpr uint32_t HALVES = kDenseBlocks / (64u / (kSyncPts + 1));
  for (uint32_t h = 0; h < HALVES; h++) {  // (a loop, not two inlined bodies)
    const uint32_t wg = g16 * HALVES + h;
    if (wg * (64u / (kSyncPts + 1)) < A.n_blocks) decode_blocks_body<0>(A, wg, stage);
    __syncthreads();
  }
}
// one list (b.n_blocks = 0) or two: workgroup -> (list, group of sixteen blocks)
__global__ __launch_bounds__(64) void decode_dense_kernel(DecodeArgs a, DecodeArgs b, uint32_t wgs_a, uint32_t stage_cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
  // (a branch per list: a reference chosen between the two argument blocks would be a select between their addresses -- scratch)
  if (blockIdx.x < wgs_a) decode_dense_body(a, blockIdx.x, stage, stage_cap);
  else decode_dense_body(b, blockIdx.x - wgs_a, stage, stage_cap);
}

// One WAVEFRONT per block for the two record kinds whose boundaries need no parse from the block start: a varint delta
// ends at the byte without the continuation bit (DocIdsOnly: 1000 records per block, so lane-per-block left 5 000 lanes
// for a 5 M-entry list), a raw delta is four bytes.  Lane i takes byte i of a 64-byte chunk; a lane on a terminator
// byte assembles its varint from the up to four continuation bytes before it, the record index is the count of
// terminators before it (ballot), the doc id the running sum of the deltas (wavefront scan): ids are written in order,
// coalesced.  Bit-for-bit the sequential decoder's output (reference inverted_index/src/codec/doc_ids_only.rs:38-49,
// raw_doc_ids_only.rs; varint/src/lib.rs).
template <int KIND>
__global__ __launch_bounds__(256) void decode_blocks_wave_kernel(const uint8_t *__restrict__ bytes,
                                                                const uint64_t *__restrict__ byte_off,
                                                                const uint32_t *__restrict__ first,
                                                                const uint32_t *__restrict__ nent,
                                                                const uint32_t *__restrict__ entry_off, uint32_t n_blocks,
                                                                uint32_t *__restrict__ ids, uint32_t *__restrict__ freqs,
                                                                uint32_t *__restrict__ masks) {
  const uint32_t lane = threadIdx.x & 63u, b = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (b >= n_blocks) return;
  const uint64_t beg = byte_off[b];
  const uint32_t len = (uint32_t)(byte_off[b + 1] - beg), n = nent[b], f0 = first[b], out0 = entry_off[b];
  const uint8_t *__restrict__ p = bytes + beg;
  if (KIND == 2) {  // raw: id = block's first doc id + u32
    for (uint32_t e = lane; e < n && 4u * e + 4u <= len; e += 64u) {
      const uint32_t d = (uint32_t)p[4 * e] | ((uint32_t)p[4 * e + 1] << 8) | ((uint32_t)p[4 * e + 2] << 16) |
                         ((uint32_t)p[4 * e + 3] << 24);
      ids[out0 + e] = f0 + d;
      if (freqs) freqs[out0 + e] = 0;
      if (masks) masks[out0 + e] = 0;
    }
    return;
  }
  uint32_t base = f0, done = 0;  // wave-uniform: doc id and number of the records emitted so far
  for (uint32_t c0 = 0; c0 < len && done < n; c0 += 64u) {
    const uint32_t i = c0 + lane;
    const bool in = i < len;
    const uint32_t c = in ? p[i] : 0x80u;
    const bool term = in && !(c & 0x80u);
    uint32_t val = 0;
    if (term) {
      // continuation bytes right before this one (at most four for a 32-bit value), oldest first
      uint32_t k = 0;
      uint32_t pb[4];
#pragma unroll
      for (uint32_t j = 1; j <= 4; j++) {
        pb[j - 1] = (i >= j && k == j - 1) ? p[i - j] : 0u;
        if (k == j - 1 && i >= j && (pb[j - 1] & 0x80u)) k = j;
      }
      // val = c_first & 0x7f; then val = ((val + 1) << 7) | (next & 0x7f) for every following byte
      bool open = false;
#pragma unroll
      for (int j = 3; j >= 0; j--)
        if ((uint32_t)j < k) {
          val = open ? (((val + 1u) << 7) | (pb[j] & 0x7fu)) : (pb[j] & 0x7fu);
          open = true;
        }
      val = open ? (((val + 1u) << 7) | (c & 0x7fu)) : (c & 0x7fu);
    }
    const unsigned long long tm = __ballot(term);
    const uint32_t before = (uint32_t)__popcll(tm & ((1ull << lane) - 1ull));
    // inclusive scan of the deltas over the wavefront (non-terminators carry 0)
    uint32_t sum = val;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
      const uint32_t o = __shfl_up(sum, d);
      if (lane >= d) sum += o;
    }
    const uint32_t idx = done + before;
    if (term && idx < n) {
      ids[out0 + idx] = base + sum;
      if (freqs) freqs[out0 + idx] = 0;
      if (masks) masks[out0 + idx] = 0;
    }
    base += __shfl(sum, 63);
    done += (uint32_t)__popcll(tm);
  }
}

// ---- intersection ----------------------------------------------------------------------------------
// (lower_bound, wave_lower_bound_range, shared_id, to_list_frame: postings_ops.hpp)

// Intersection probe.  The 256 candidates of a workgroup are consecutive in the (sorted) driving list, so their
// matches in another list lie in ONE window [lower_bound(first), lower_bound(first of the next workgroup)): the
// window's ends are found by wavefront-wide 64-ary searches, the window is staged in LDS with coalesced loads and
// every lane finishes with a binary search in LDS.  Compared with 2.5 M independent 23-step searches through L2
// this reads each list once, sequentially.  Windows that do not fit (very skewed lists) fall back to a per-lane
// binary search confined to the window.
constexpr uint32_t kProbeWindow = 4096;  // u32 entries of the other list staged per workgroup (16 KiB)

// DPT: drivers per thread.  A workgroup's chain is four dependent global round trips (the window's ends) + the staging + an
// LDS search, and 2.5 M drivers in tiles of 256 are 9 766 such chains in ~5 rounds over the CUs (27 us, configs[4]); tiles of
// 1 024 (DPT = 4, window of 8 Ki entries) amortise the chain over four times the drivers.
template <int DPT>
__global__ __launch_bounds__(256) void intersect_probe_kernel(ListView v, uint8_t *__restrict__ flags,
                                                              uint32_t *__restrict__ pos,
                                                              uint32_t *__restrict__ block_counts) {
  constexpr uint32_t TILE = 256 * DPT, WIN = kProbeWindow * (DPT > 1 ? 2 : 1);
  __shared__ uint32_t win[WIN];
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t w_lo, w_hi;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n0 = v.len[0];
  const uint32_t i_first = blockIdx.x * TILE, i_next = i_first + TILE;
  // (end of round 3, from hybrid_tile_kernel: the first level of the window-end searches in list 1 -- 64 positions that depend on
  // the list's length alone -- is requested together with the tile's doc ids; the searches stop one level early (a window up
  // to 64 entries wider at either end instead of one more dependent round trip); the window is staged with up to eight loads
  // per lane in flight; a lane's searches in LDS are fixed-length and advance in step)
  uint32_t lvl1 = 0;
  const bool pre1 = v.n > 1 && v.len[1] > 64;
  if (pre1) {
    const uint32_t step = (v.len[1] + 63) / 64, p = (lane + 1) * step - 1;
    lvl1 = v.ids[1][p < v.len[1] ? p : v.len[1] - 1];
  }
  bool hit[DPT];
  uint32_t xc[DPT];  // shared frame
#pragma unroll
  for (int k = 0; k < DPT; k++) {
    const uint32_t i = i_first + k * 256 + threadIdx.x;
    hit[k] = i < n0;
    xc[k] = shared_id(v, 0, hit[k] ? i : n0 - 1);
  }
  const uint32_t x_first = shared_id(v, 0, i_first), x_next = shared_id(v, 0, i_next < n0 ? i_next : n0 - 1);
  for (int l = 1; l < v.n; l++) {
    const uint32_t *__restrict__ a = v.ids[l];
    const uint32_t nl = v.len[l];
    if (wave == 0) {  // the window's start: at or below lower_bound(first driver)
      bool u0;
      uint32_t rlo, rhi;
      wave_lower_bound_range(a, nl, to_list_frame(x_first, v.add[l], &u0), lane, lvl1, l == 1 && pre1, &rlo, &rhi);
      if (lane == 0) w_lo = rlo;
    } else if (wave == 1) {  // its end: at or above lower_bound(first driver of the next tile)
      bool u1;
      uint32_t rlo, rhi = nl;
      if (i_next < n0) wave_lower_bound_range(a, nl, to_list_frame(x_next, v.add[l], &u1), lane, lvl1, l == 1 && pre1, &rlo, &rhi);
      if (lane == 0) w_hi = rhi;
    }
    __syncthreads();
    const uint32_t lo = w_lo;
    const uint32_t hi = w_hi;  // every candidate x of this workgroup has lower_bound(x) in [lo, hi]
    const uint32_t span = hi - lo;
    if (span <= WIN) {
      for (uint32_t base = 0; base < span; base += 8 * 256) {
        uint32_t t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t o = base + j * 256 + threadIdx.x;
          t[j] = a[lo + (o < span ? o : span - 1)];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t o = base + j * 256 + threadIdx.x;
          if (o < span) win[o] = t[j];
        }
      }
      __syncthreads();
      uint32_t xl[DPT], b[DPT];
      bool under[DPT];
#pragma unroll
      for (int k = 0; k < DPT; k++) {
        xl[k] = to_list_frame(xc[k], v.add[l], &under[k]);  // this list's frame
        b[k] = 0;
      }
      uint32_t rem = span;
      while (rem > 1) {
        const uint32_t half = rem >> 1;
#pragma unroll
        for (int k = 0; k < DPT; k++) b[k] = win[b[k] + half - 1] < xl[k] ? b[k] + half : b[k];
        rem -= half;
      }
#pragma unroll
      for (int k = 0; k < DPT; k++) {
        const uint32_t i = i_first + k * 256 + threadIdx.x;
        if (span && win[b[k]] < xl[k]) b[k]++;
        const bool m = hit[k] && !under[k] && b[k] < span && win[b[k] < span ? b[k] : 0] == xl[k];
        if (hit[k]) pos[(size_t)(l - 1) * n0 + i] = lo + b[k];
        hit[k] = m;
      }
    } else {
#pragma unroll
      for (int k = 0; k < DPT; k++) {
        const uint32_t i = i_first + k * 256 + threadIdx.x;
        bool under;
        const uint32_t x = to_list_frame(xc[k], v.add[l], &under);
        uint32_t b = lo, e = hi;
        if (hit[k]) {
          while (b < e) {
            const uint32_t mid = b + ((e - b) >> 1);
            if (a[mid] < x) b = mid + 1;
            else e = mid;
          }
          pos[(size_t)(l - 1) * n0 + i] = b;
          hit[k] = !under && b < nl && a[b] == x;
        }
      }
    }
    __syncthreads();  // win / w_lo / w_hi are reused by the next list
  }
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < DPT; k++) {
    const uint32_t i = i_first + k * 256 + threadIdx.x;
    if (i < n0) flags[i] = hit[k] ? 1 : 0;
    cnt += (uint32_t)__popcll(__ballot(hit[k]));
  }
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// single workgroup: exclusive scan of nb counters.  1024 threads x 8 consecutive counters per step (two 16-byte loads
// each, all in flight before the first use), thread-local scan, shuffle scan across the wave, 16 wave totals through LDS.
__global__ __launch_bounds__(1024) void scan_counts_kernel(uint32_t *__restrict__ c, uint32_t nb,
                                                           uint32_t *__restrict__ total_out) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 8192) {
    const uint32_t i0 = base + threadIdx.x * 8;
    uint32_t x[8];
    if (i0 + 8 <= nb && (reinterpret_cast<uintptr_t>(c + i0) & 15u) == 0) {
      const uint4 a = *reinterpret_cast<const uint4 *>(c + i0), b = *reinterpret_cast<const uint4 *>(c + i0 + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = i0 + j < nb ? c[i0 + j] : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) sum += x[j];
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= (uint32_t)off) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t j = 0; j < w; j++) woff += wsum[j];
    uint32_t run = carry + woff + inc - sum;  // exclusive prefix of this thread's first counter
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (i0 + j < nb) c[i0 + j] = run;
      run += x[j];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = run;
    __syncthreads();
  }
  if (threadIdx.x == 0) total_out[0] = carry;
}

template <int DPT>
__global__ __launch_bounds__(256) void intersect_write_kernel(ListView v, LeafMap lm, const uint8_t *__restrict__ flags,
                                                              const uint32_t *__restrict__ pos,
                                                              const uint32_t *__restrict__ block_off,
                                                              uint32_t *__restrict__ out_ids,
                                                              uint32_t *__restrict__ out_freqs, uint32_t cap,
                                                              uint32_t *__restrict__ out_epos) {
  constexpr uint32_t TILE = 256 * DPT;  // (the probe's tile: block_off holds one exclusive prefix per tile)
  __shared__ uint32_t wave_cnt[DPT][4];
  const uint32_t n0 = v.len[0];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  bool hit[DPT];
  unsigned long long m[DPT];
#pragma unroll
  for (int k = 0; k < DPT; k++) {
    const uint32_t i = blockIdx.x * TILE + k * 256 + threadIdx.x;
    hit[k] = i < n0 && flags[i];
    m[k] = __ballot(hit[k]);
    if (lane == 0) wave_cnt[k][w] = (uint32_t)__popcll(m[k]);
  }
  __syncthreads();
  uint32_t run = block_off[blockIdx.x];  // hits before slice k of this tile
#pragma unroll
  for (int k = 0; k < DPT; k++) {
    if (hit[k]) {
      const uint32_t i = blockIdx.x * TILE + k * 256 + threadIdx.x;
      uint32_t off = run;
      for (uint32_t j = 0; j < w; j++) off += wave_cnt[k][j];
      off += (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull));
      out_ids[off] = shared_id(v, 0, i);
      for (int l = 0; l < lm.n_leaves; l++) {
        const uint32_t t = lm.leaf_list[l];
        const uint32_t p = t == 0 ? i : pos[(size_t)(t - 1) * n0 + i];  // the hit's position in list t
        // (a codec that stores no frequency yields the term record's default, 1: reference index_result/src/core/mod.rs:192-197)
        if (out_freqs) out_freqs[(size_t)l * cap + off] = lm.leaf_freq[l] ? lm.leaf_freq[l][p] : 1u;
        if (out_epos) out_epos[(size_t)l * cap + off] = lm.leaf_epos[l] ? lm.leaf_epos[l][p] : p;
      }
    }
    run += wave_cnt[k][0] + wave_cnt[k][1] + wave_cnt[k][2] + wave_cnt[k][3];
  }
}

// ---- union / NOT --------------------------------------------------------------------------------------
// Union of sorted lists without a sort: an element of list s is EMITTED by s iff no earlier list holds it
// (union_flag); with the exclusive prefix counts of the emitted flags of every list (union_prefix), the
// output slot of an emitted element x is  sum over lists t of  prefix_t[lower_bound_t(x)]  -- the number of
// emitted elements smaller than x -- so every list writes its own elements straight to their final, doc-id
// ordered position (union_write).  Reference: rqe_iterators/src/union_flat.rs:223-257,297-320.
__global__ __launch_bounds__(256) void union_flag_kernel(ListView v, int s, uint8_t *__restrict__ flags,
                                                         uint32_t *__restrict__ block_counts) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  bool emit = i < v.len[s];
  if (emit) {
    const uint32_t xc = shared_id(v, s, i);
    for (int t = 0; t < s; t++) {
      bool under;
      const uint32_t x = to_list_frame(xc, v.add[t], &under);
      if (under) continue;
      const uint32_t p = lower_bound(v.ids[t], v.len[t], x);
      if (p < v.len[t] && v.ids[t][p] == x) { emit = false; break; }
    }
    flags[i] = emit ? 1 : 0;
  }
  unsigned long long m = __ballot(emit);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// prefix[i] = emitted elements before i, i in [0, len]; block_off = exclusive scan of the block counts
__global__ __launch_bounds__(256) void union_prefix_kernel(const uint8_t *__restrict__ flags, uint32_t len,
                                                           const uint32_t *__restrict__ block_off,
                                                           const uint32_t *__restrict__ total,
                                                           uint32_t *__restrict__ prefix) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool f = i < len && flags[i];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long m = __ballot(f);
  if (lane == 0) wave_cnt[w] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = block_off[blockIdx.x];
  for (uint32_t j = 0; j < w; j++) off += wave_cnt[j];
  off += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (i < len) prefix[i] = off;
  if (i == len) prefix[len] = total[0];  // (the grid covers len + 1 slots)
}

__global__ __launch_bounds__(256) void union_write_kernel(ListView v, LeafMap lm, UnionView u, int s,
                                                          const uint8_t *__restrict__ flags,
                                                          uint32_t *__restrict__ out_ids,
                                                          uint32_t *__restrict__ out_freqs, uint32_t cap,
                                                          uint32_t *__restrict__ out_epos) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= v.len[s] || !flags[i]) return;
  const uint32_t xc = shared_id(v, s, i);
  uint32_t slot = 0, at[kMaxLists];
  bool outside[kMaxLists];
  for (int t = 0; t < v.n; t++) {
    const uint32_t x = to_list_frame(xc, v.add[t], &outside[t]);
    // (outside above the list's range: every entry is smaller, unless one sits exactly at the clamp -- lower_bound
    // would stop in front of it, so the end is taken explicitly)
    at[t] = t == s ? i : (outside[t] ? (x ? v.len[t] : 0u) : lower_bound(v.ids[t], v.len[t], x));
    slot += u.prefix[t][at[t]];
  }
  if (slot >= cap) return;
  out_ids[slot] = xc;
  for (int l = 0; l < lm.n_leaves; l++) {
    const uint32_t t = lm.leaf_list[l], p = at[t];
    const bool on = p < v.len[t] && !outside[t] && shared_id(v, (int)t, p) == xc;
    out_freqs[(size_t)l * cap + slot] = on ? (lm.leaf_freq[l] ? lm.leaf_freq[l][p] : 1u) : 0u;
    if (out_epos) out_epos[(size_t)l * cap + slot] = on ? (lm.leaf_epos[l] ? lm.leaf_epos[l][p] : p) : 0xFFFFFFFFu;
  }
}

// NOT (rqe_iterators/src/not.rs:171-209; not_optimized.rs with a universe list): candidate c is doc id c+1
// (no universe) or universe[c]; it survives iff the child does not hold it, and then its slot is
// c - (#child entries below it that are candidates themselves).  Without a universe every child entry
// <= max_doc is a candidate, so the slot is c - lower_bound(child, doc).  With a universe the count of
// excluded predecessors comes from the same flag / scan / write triple as the intersection.
__global__ __launch_bounds__(256) void not_range_kernel(const uint32_t *__restrict__ child, uint32_t child_len,
                                                        uint32_t max_doc, uint32_t *__restrict__ out_ids,
                                                        uint32_t *__restrict__ out_freqs, uint32_t cap) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= max_doc) return;
  const uint32_t doc = c + 1;
  const uint32_t p = lower_bound(child, child_len, doc);
  if (p < child_len && child[p] == doc) return;
  const uint32_t slot = c - p;
  if (slot < cap) {
    out_ids[slot] = doc;
    out_freqs[slot] = 1;  // virtual result: scored as idf = f = 1 (reference src/ext/default.c:289-293)
  }
}
__global__ void count_below_kernel(const uint32_t *__restrict__ list, uint32_t len, uint64_t x, uint32_t *out) {
  out[0] = x > 0xFFFFFFFFull ? len : lower_bound(list, len, (uint32_t)x);
}
__global__ __launch_bounds__(256) void not_universe_flag_kernel(const uint32_t *__restrict__ universe, uint32_t n_u,
                                                                const uint32_t *__restrict__ child, uint32_t child_len,
                                                                long long child_shift, uint32_t max_doc,
                                                                uint8_t *__restrict__ flags,
                                                                uint32_t *__restrict__ block_counts) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  bool keep = i < n_u;
  if (keep) {
    const uint32_t doc = universe[i];
    bool outside;  // the child stores ids relative to its own base: child frame = universe frame + child_shift
    const uint32_t cdoc = to_list_frame(doc, -child_shift, &outside);
    const uint32_t p = outside ? child_len : lower_bound(child, child_len, cdoc);
    keep = doc <= max_doc && !(p < child_len && child[p] == cdoc);
    flags[i] = keep ? 1 : 0;
  }
  unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}
__global__ __launch_bounds__(256) void not_universe_write_kernel(const uint32_t *__restrict__ universe, uint32_t n_u,
                                                                 const uint8_t *__restrict__ flags,
                                                                 const uint32_t *__restrict__ block_off,
                                                                 uint32_t *__restrict__ out_ids,
                                                                 uint32_t *__restrict__ out_freqs, uint32_t cap) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool keep = i < n_u && flags[i];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long m = __ballot(keep);
  if (lane == 0) wave_cnt[w] = (uint32_t)__popcll(m);
  __syncthreads();
  if (!keep) return;
  uint32_t off = block_off[blockIdx.x];
  for (uint32_t j = 0; j < w; j++) off += wave_cnt[j];
  off += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (off < cap) {
    out_ids[off] = universe[i];
    out_freqs[off] = 1;
  }
}

// ---- proximity: the per-candidate / per-hit device functions live in postings_ops.hpp (hybrid_kernels.hip runs them too) ----
template <int MAXL>
__global__ __launch_bounds__(256) void prox_filter_kernel(ProxParams P, OffsetView o, LeafMap lm, uint32_t n0,
                                                          const uint32_t *__restrict__ pos, uint8_t *__restrict__ flags,
                                                          uint32_t *__restrict__ block_counts) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  bool keep = i < n0 && flags[i];
  if (keep) {
    auto entry = [&](int l) {
      const uint32_t t = lm.leaf_list[l];
      const uint32_t p = t == 0 ? i : pos[(size_t)(t - 1) * n0 + i];
      return lm.leaf_epos[l] ? lm.leaf_epos[l][p] : p;
    };
    if (prox_two_terms(P)) {  // (two plain terms: cursors in registers, postings_ops.hpp)
      keep = prox_within_range2(P, prox_term(o, 0, entry(0)), prox_term(o, 1, entry(1)));
    } else {
      ProxCtx<MAXL> x;
      prox_load<MAXL>(P, o, x, entry);
      keep = prox_within_range<MAXL>(P, x);
    }
    flags[i] = keep ? 1 : 0;
  }
  unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

template <int MAXL>
__global__ __launch_bounds__(256) void prox_slop_kernel(ProxParams P, OffsetView o, const uint32_t *__restrict__ epos,
                                                        uint32_t len, uint32_t cap, int32_t *__restrict__ slops) {
  const uint32_t h = blockIdx.x * 256 + threadIdx.x;
  if (h >= len) return;
  if (prox_two_terms(P) && !P.count_present) {
    slops[h] = prox_min_offset_delta2(prox_term(o, 0, epos[h]), prox_term(o, 1, epos[(size_t)cap + h]));
    return;
  }
  ProxCtx<MAXL> x;
  prox_load<MAXL>(P, o, x, [&](int l) { return epos[(size_t)l * cap + h]; });
  slops[h] = prox_min_offset_delta<MAXL>(P, x);
}

// ---- scorers ---------------------------------------------------------------------------------------
// DEEP: the tree is deeper than root -> groups -> leaves (P.n_nodes > 0).  A kernel of its own because its per-level
// accumulators are indexed dynamically and live in scratch memory: the flat / two-level kernel must not pay for them.
template <bool DEEP>
__global__ __launch_bounds__(256) void score_kernel(ScoreParams P, const uint32_t *__restrict__ ids,
                                                    const uint32_t *__restrict__ freqs, uint32_t len, uint32_t cap,
                                                    const uint32_t *__restrict__ doc_len,
                                                    const float *__restrict__ doc_score,
                                                    const uint32_t *__restrict__ max_freq, uint32_t table_n,
                                                    double *__restrict__ scores, uint64_t *__restrict__ keys,
                                                    uint32_t *__restrict__ keys32) {
  const uint32_t h = blockIdx.x * 256 + threadIdx.x;
  if (h >= len) return;
  const long long tid = (long long)ids[h] + P.table_off;
  const bool known = tid >= 0 && tid < (long long)table_n;
  const uint32_t id = known ? (uint32_t)tid : 0u;
  const float dscore = known ? doc_score[id] : 0.0f;
  const uint32_t dlen = known ? doc_len[id] : 0u;
  const uint32_t mfreq = (known && max_freq) ? max_freq[id] : 0u;
  auto F = [&](int t) { return (double)freqs[(size_t)t * cap + h]; };
  // IndexResult_MinOffsetDelta of offset-less children = (children in the aggregate) - 1, at least 1
  // (reference src/index_result/index_result.c:51-103); a union's aggregate only holds the children that
  // matched this document (union_flat.rs:297-320)
  int slop = P.slops ? P.slops[h] : P.slop;
  if (P.is_union && !P.slops) {
    int matched = 0;
    for (int g = 0; g < P.n_groups; g++) {
      bool any = false;
      for (int t = P.group_first[g]; t < P.group_first[g + 1]; t++) any |= freqs[(size_t)t * cap + h] != 0;
      matched += any ? 1 : 0;
    }
    slop = matched > 1 ? matched - 1 : 1;
  }
  const double s = score_one<DEEP>(P, F, dlen, dscore, mfreq, slop);
  scores[h] = s;
  if (keys) keys[h] = ~d2key(s);  // descending score; the select's row tie-break = ascending doc id
  if (keys32) {
    // 32-bit PREFILTER key: the orderable image of (float)(-score).  double -> float rounding is monotone, so
    // score a > score b  =>  keys32[a] <= keys32[b]: a threshold pass over these 4-byte keys keeps a superset of the
    // top-N (ties included), the exact order is then settled on the 64-bit keys of the few survivors.
    const uint32_t u = __float_as_uint((float)(-s));
    keys32[h] = ((u & 0x7fffffffu) > 0x7f800000u) ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
  }
}

// survivors of the prefilter -> pinned host memory: row (= hit index) and full 64-bit key of the first
// min(count, cap) candidates, and the count itself
__global__ __launch_bounds__(256) void fetch_cand64_kernel(const uint2 *__restrict__ cand, const uint32_t *__restrict__ count,
                                                           uint32_t cap, const uint64_t *__restrict__ keys64,
                                                           const uint32_t *__restrict__ ids,
                                                           uint32_t *__restrict__ out_rows, uint64_t *__restrict__ out_keys,
                                                           uint32_t *__restrict__ out_ids, uint32_t *__restrict__ out_n) {
  const uint32_t n = count[0];
  const uint32_t m = n < cap ? n : cap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
    const uint32_t row = cand[i].x;
    out_rows[i] = row;
    out_keys[i] = keys64[row];
    out_ids[i] = ids[row];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out_n[0] = n;
}

// ---- BM25STD.NORM: score / max score over ALL hits (reference RPMaxScoreNormalizer, src/result_processor.c:1770-1812:
// maxValue starts at 0, MAX() over every upstream score, division only when maxValue != 0) ----------------------------
__device__ __forceinline__ double key2d(uint64_t k) {
  const uint64_t u = (k & 0x8000000000000000ull) ? (k ^ 0x8000000000000000ull) : ~k;
  return __longlong_as_double((long long)u);
}
// max_key[0] (zeroed by the caller) <- max over the orderable images of the scores; one atomic per wavefront
__global__ __launch_bounds__(256) void score_max_kernel(const double *__restrict__ scores, uint32_t n,
                                                        unsigned long long *__restrict__ max_key) {
  unsigned long long m = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const unsigned long long k = d2key(scores[i]);
    m = k > m ? k : m;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(m, off, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(max_key, m);
}
__global__ __launch_bounds__(256) void score_normalize_kernel(double *__restrict__ scores, uint64_t *__restrict__ keys,
                                                              uint32_t n, const unsigned long long *__restrict__ max_key) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long zero = 0x8000000000000000ull;  // d2key(0.0): the accumulator's start value
  const unsigned long long mk = max_key[0] > zero ? max_key[0] : zero;
  const double mx = key2d(mk);
  if (mx != 0.0) {
    const double s = scores[i] / mx;
    scores[i] = s;
    if (keys) keys[i] = ~d2key(s);
  }
}

__global__ __launch_bounds__(256) void labels_to_rows_kernel(const uint32_t *__restrict__ ids, uint32_t n,
                                                             uint64_t ids_base, LabelRows L, uint32_t *__restrict__ rows) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  rows[i] = label_first_row(L, ids_base + ids[i]);
}

// The hits that have a vector, compacted (order does not matter: the selection that follows ranks by (distance, hit
// index)): rows_out[slot] = storage row, cand[slot] = (hit index, 0); count[0] = how many (slots >= cap are dropped and
// the caller sees count > cap).  One atomic per wavefront.
__global__ __launch_bounds__(256) void labels_to_cand_kernel(const uint32_t *__restrict__ ids, uint32_t n, uint64_t ids_base,
                                                             LabelRows L, uint32_t *__restrict__ rows_out,
                                                             uint2 *__restrict__ cand, uint32_t *__restrict__ count,
                                                             uint32_t cap) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  uint32_t row = kNoRow;
  if (i < n) row = label_first_row(L, ids_base + ids[i]);
  const bool has = row != kNoRow;
  const unsigned long long m = __ballot(has);
  if (!m) return;
  uint32_t first = 0;
  if (lane == (uint32_t)__builtin_ctzll(m)) first = atomicAdd(count, (uint32_t)__popcll(m));
  first = __shfl(first, __builtin_ctzll(m), 64);
  if (!has) return;
  const uint32_t slot = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (slot < cap) {
    rows_out[slot] = row;
    cand[slot] = make_uint2(i, 0u);
  }
}

// k (<= 32) nearest of the compacted candidates in ONE launch: block b ranks its share of the candidates by the composite
// (distance key, hit index) with k rounds of a block-wide arg-min and parks its k best in `part`; the block that finishes
// last (a ticket) ranks the blocks' lists the same way and writes the winners -- hit index, key, doc id -- into pinned
// host memory.  It also puts the two counters back to zero, so the stream needs no memset before the next query.
// (wave_min_u64 / wave_topk and the constants: postings_ops.hpp)
__global__ __launch_bounds__(256) void knn_topk_kernel(const float *__restrict__ dists, const uint2 *__restrict__ cand,
                                                       uint32_t *__restrict__ count, uint32_t cap, uint32_t k,
                                                       const uint32_t *__restrict__ ids, uint64_t *__restrict__ part,
                                                       uint32_t *__restrict__ ticket, uint32_t *__restrict__ out_rows,
                                                       uint32_t *__restrict__ out_keys, uint32_t *__restrict__ out_ids,
                                                       uint32_t *__restrict__ out_n, uint32_t *__restrict__ overflow) {
  __shared__ uint64_t wl[4][kKnnTopkMaxK];  // the four waves' lists
  __shared__ uint32_t sh_last;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t total = *count;
  const uint32_t n = total < cap ? total : cap;
  const uint32_t per = (n + kKnnTopkBlocks - 1) / kKnnTopkBlocks;  // <= 256 * kKnnTopkPerThread (cap = 64 Ki)
  const uint32_t beg = blockIdx.x * per, end = beg + per < n ? beg + per : n;
  uint64_t mine[kKnnTopkPerThread];
#pragma unroll
  for (int j = 0; j < kKnnTopkPerThread; j++) {
    const uint32_t i = beg + j * 256 + threadIdx.x;
    mine[j] = i < end ? (((uint64_t)f2key(dists[i]) << 32) | cand[i].x) : ~0ull;
  }
  {
    const uint64_t r = wave_topk(mine, k, lane);
    if (lane < kKnnTopkMaxK) wl[w][lane] = r;
  }
  __syncthreads();
  if (w == 0) {  // merge the four lists: 4 * k <= 128 composites, two per lane
    uint64_t two[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t i = j * 64 + lane;
      two[j] = (i % kKnnTopkMaxK) < k ? wl[i / kKnnTopkMaxK][i % kKnnTopkMaxK] : ~0ull;
    }
    const uint64_t r = wave_topk(two, k, lane);
    if (lane < kKnnTopkMaxK) part[(size_t)blockIdx.x * kKnnTopkMaxK + lane] = r;
    __threadfence();
    if (lane == 0) sh_last = atomicAdd(ticket, 1u) == (uint32_t)kKnnTopkBlocks - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!sh_last) return;
  __threadfence();
  // the last block: the kKnnTopkBlocks lists (<= 2048 composites, 8 per thread), per wave first, then merged
  constexpr int PT = kKnnTopkBlocks * kKnnTopkMaxK / 256;
  uint64_t all[PT];
#pragma unroll
  for (int j = 0; j < PT; j++) {
    const uint32_t i = j * 256 + threadIdx.x, t = i % kKnnTopkMaxK;
    all[j] = t < k ? part[i] : ~0ull;
  }
  {
    const uint64_t r = wave_topk(all, k, lane);
    if (lane < kKnnTopkMaxK) wl[w][lane] = r;  // (wave 0 finished reading wl before the barrier above)
  }
  __syncthreads();
  if (w != 0) return;
  uint64_t two[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const uint32_t i = j * 64 + lane;
    two[j] = (i % kKnnTopkMaxK) < k ? wl[i / kKnnTopkMaxK][i % kKnnTopkMaxK] : ~0ull;
  }
  const uint64_t m = wave_topk(two, k, lane);
  uint32_t got = 0;
  if (lane < k && m != ~0ull) {
    const uint32_t hit = (uint32_t)m;
    out_rows[lane] = hit;
    out_keys[lane] = (uint32_t)(m >> 32);
    out_ids[lane] = ids[hit];
    got = 1;
  }
  const uint32_t n_got = (uint32_t)__popcll(__ballot(got != 0));  // the valid entries are a prefix (ascending, ~0 last)
  if (lane == 0) {
    *out_n = n_got;
    if (total > cap) *overflow = 1;  // more candidates than the list holds: the caller redoes the query the long way
    *count = 0;
    *ticket = 0;
  }
}

__global__ __launch_bounds__(256) void dist_to_keys_kernel(const float *__restrict__ d, uint32_t n,
                                                           uint32_t *__restrict__ keys) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t u = __float_as_uint(d[i]);
  keys[i] = ((u & 0x7fffffffu) > 0x7f800000u) ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}

// out[i] = src[idx[i]] (idx / out may be host-visible pinned memory)
__global__ __launch_bounds__(256) void gather_u32_counted_kernel(const uint32_t *__restrict__ src, uint32_t src_len,
                                                                 const uint32_t *__restrict__ idx,
                                                                 const uint32_t *__restrict__ count, uint32_t cap,
                                                                 uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t n = *count < cap ? *count : cap;
  if (i >= n) return;
  const uint32_t j = idx[i];
  if (j < src_len) out[i] = src[j];
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                                         uint32_t n, uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

inline uint32_t blocks_for(uint32_t n) { return n ? (n + 255) / 256 : 1; }

}  // namespace

bool decode_sync_supported(const CodecDesc &cd) { return cd.kind == 0 && !cd.wide; }
size_t decode_sync_words(uint32_t n_blocks) { return (size_t)n_blocks * kSyncPts * 2; }
uint32_t decode_sync_blocks_per_wave() { return 64 / (kSyncPts + 1); }
uint32_t decode_stage_bytes() { return kDecodeLds; }

// sync_mode after the rules of the kernel + the staging size that goes with it
static int decode_mode(const CodecDesc &cd, const uint32_t *sync, const uint32_t *wmasks, int sync_mode, uint32_t sync_span,
                       uint32_t *lds_cap) {
  if (!sync || !decode_sync_supported(cd) || wmasks) sync_mode = 0;
  // sync_mode 2: eight lanes per block.  Only layouts the staged fast parsers take may use it.  Lists with inline offsets
  // (Full, *Offsets: round 4) too -- their sync points are only allocated when every 8-block span fits the staging buffer
  // (RSGPU_Postings_Upload); a wavefront that does not fit parses whole blocks, one lane each (decode_blocks_body).
  // staging bytes: everything a wavefront of sync_mode 2 can need (the caller knows the widest 8-block span), else 30 KiB
  *lds_cap = sync_mode == 2 && sync_span && sync_span < kDecodeLds ? ((sync_span + 255u) & ~255u) : kDecodeLds;
  return sync_mode;
}

// blocks per wavefront of the lane-per-block modes: as many (a power of two, <= 64) as keep the wavefront's byte range inside
// the staging buffer for blocks of the list's average length + 30 % (FreqsOnly: ~300 B -> 64; Full: ~800 B -> 16 or 32)
static uint32_t decode_bpw1(uint32_t avg_block_bytes) {
  uint32_t bpw = 64;
  while (bpw > 4 && (uint64_t)bpw * avg_block_bytes * 13 / 10 + 64 > kDecodeLds) bpw >>= 1;
  return bpw;
}

// the dense-group kernel takes a list (or two) when its layout has no inline offsets, its sync points are there (the groups that are
// not dense go through them) and sixteen blocks fit the staging buffer
static bool dense_eligible(const DecodeArgs &a) {
  return scan_tuning().decode_dense && a.sync_mode == 2 && a.cd.kind == 0 && !a.cd.wide && a.cd.osz < 0 && !a.wmasks && !a.off_pos &&
         ((a.cd.n == 2 && ((a.cd.freq == 1 && a.cd.mask == -1) || (a.cd.freq == -1 && a.cd.mask == 1))) ||
          (a.cd.n == 3 && a.cd.freq == 1 && a.cd.mask == 2)) &&
         kDenseSpans * a.lds_cap + 64 <= 64 * 1024;
}
static void launch_decode_dense(const DecodeArgs &a, const DecodeArgs *b, hipStream_t s) {
  const uint32_t cap = b && b->lds_cap > a.lds_cap ? b->lds_cap : a.lds_cap;
  const uint32_t wgs_a = (a.n_blocks + kDenseBlocks - 1) / kDenseBlocks, wgs_b = b ? (b->n_blocks + kDenseBlocks - 1) / kDenseBlocks : 0;
  DecodeArgs none = a;
  none.n_blocks = 0;
  // (the staging buffer: the group's eight-block spans + the parsers' slack)
  hipLaunchKernelGGL(decode_dense_kernel, dim3(wgs_a + wgs_b), dim3(64), kDenseSpans * cap + 64, s, a, b ? *b : none, wgs_a, kDenseSpans * cap);
}

void launch_decode_blocks(const CodecDesc &cd, const uint8_t *bytes, const uint64_t *byte_off, const uint32_t *first,
                          const uint32_t *nent, const uint32_t *entry_off, uint32_t n_blocks, uint32_t *ids,
                          uint32_t *freqs, uint32_t *masks, hipStream_t s, uint32_t *wmasks, uint32_t *off_pos,
                          uint32_t *off_len, uint32_t *sync, int sync_mode, uint32_t sync_span, uint32_t avg_block_bytes) {
  if (!n_blocks) return;
  uint32_t lds_cap;
  sync_mode = decode_mode(cd, sync, wmasks, sync_mode, sync_span, &lds_cap);
  const uint32_t bpw1 = decode_bpw1(avg_block_bytes);
  const uint32_t bpw = sync_mode == 2 ? 64 / (kSyncPts + 1) : bpw1;
  const DecodeArgs a{cd, bytes, byte_off, first, nent, entry_off, n_blocks, ids, freqs, masks, wmasks, off_pos, off_len, sync,
                     sync_mode, lds_cap, bpw1};
  if (dense_eligible(a)) {
    launch_decode_dense(a, nullptr, s);
    return;
  }
#define RSGPU_DECODE(K) hipLaunchKernelGGL(decode_blocks_kernel<K>, dim3((n_blocks + bpw - 1) / bpw), dim3(64), lds_cap + 64, s, a)
  // varint / raw deltas without a wide mask: one wavefront per block (decode_blocks_wave_kernel)
  const bool wave = (cd.kind == 1 || cd.kind == 2) && !cd.wide && !wmasks && !off_pos;
  if (wave && cd.kind == 1)
    hipLaunchKernelGGL(decode_blocks_wave_kernel<1>, dim3((n_blocks + 3) / 4), dim3(256), 0, s, bytes, byte_off, first, nent,
                       entry_off, n_blocks, ids, freqs, masks);
  else if (wave)
    hipLaunchKernelGGL(decode_blocks_wave_kernel<2>, dim3((n_blocks + 3) / 4), dim3(256), 0, s, bytes, byte_off, first, nent,
                       entry_off, n_blocks, ids, freqs, masks);
  else if (cd.kind == 0) RSGPU_DECODE(0);
  else if (cd.kind == 1) RSGPU_DECODE(1);
  else RSGPU_DECODE(2);
#undef RSGPU_DECODE
}

bool launch_decode_blocks_pair(const DecodeListArgs &x, const DecodeListArgs &y, hipStream_t s) {
  if (!x.n_blocks || !y.n_blocks || x.cd.kind != 0 || y.cd.kind != 0 || x.wmasks || y.wmasks) return false;
  uint32_t cap_x, cap_y;
  const int mx = decode_mode(x.cd, x.sync, x.wmasks, x.sync_mode, x.sync_span, &cap_x);
  const int my = decode_mode(y.cd, y.sync, y.wmasks, y.sync_mode, y.sync_span, &cap_y);
  const uint32_t bpw_x = mx == 2 ? 64 / (kSyncPts + 1) : 64, bpw_y = my == 2 ? 64 / (kSyncPts + 1) : 64;
  const uint32_t wgs_x = (x.n_blocks + bpw_x - 1) / bpw_x, wgs_y = (y.n_blocks + bpw_y - 1) / bpw_y;
  const DecodeArgs a{x.cd, x.bytes, x.byte_off, x.first, x.nent, x.entry_off, x.n_blocks, x.ids, x.freqs, x.masks, nullptr,
                     x.off_pos, x.off_len, x.sync, mx, cap_x, 64};
  const DecodeArgs b{y.cd, y.bytes, y.byte_off, y.first, y.nent, y.entry_off, y.n_blocks, y.ids, y.freqs, y.masks, nullptr,
                     y.off_pos, y.off_len, y.sync, my, cap_y, 64};
  if (mx != 2 || my != 2) return false;  // (the pair launch is for lists whose sync points are there)
  if (dense_eligible(a) && dense_eligible(b)) {
    launch_decode_dense(a, &b, s);
    return true;
  }
  const uint32_t lds = (cap_x > cap_y ? cap_x : cap_y) + 64;
  hipLaunchKernelGGL(decode_blocks_pair_kernel<0>, dim3(wgs_x + wgs_y), dim3(64), lds, s, a, b, wgs_x);
  return true;
}
void launch_intersect_probe(const ListView &v, uint8_t *flags, uint32_t *pos, uint32_t *block_counts, hipStream_t s, int dpt) {
  if (dpt == 4)
    hipLaunchKernelGGL(intersect_probe_kernel<4>, dim3((v.len[0] + 1023) / 1024), dim3(256), 0, s, v, flags, pos, block_counts);
  else
    hipLaunchKernelGGL(intersect_probe_kernel<1>, dim3(blocks_for(v.len[0])), dim3(256), 0, s, v, flags, pos, block_counts);
}
void launch_scan_counts(uint32_t *block_counts, uint32_t nb, uint32_t *total_out, hipStream_t s) {
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, s, block_counts, nb, total_out);
}
void launch_intersect_write(const ListView &v, const LeafMap &m, const uint8_t *flags, const uint32_t *pos,
                            const uint32_t *block_off, uint32_t *out_ids, uint32_t *out_freqs, uint32_t cap, hipStream_t s,
                            uint32_t *out_epos, int dpt) {
  if (dpt == 4)
    hipLaunchKernelGGL(intersect_write_kernel<4>, dim3((v.len[0] + 1023) / 1024), dim3(256), 0, s, v, m, flags, pos, block_off,
                       out_ids, out_freqs, cap, out_epos);
  else
    hipLaunchKernelGGL(intersect_write_kernel<1>, dim3(blocks_for(v.len[0])), dim3(256), 0, s, v, m, flags, pos, block_off,
                       out_ids, out_freqs, cap, out_epos);
}
void launch_prox_filter(const ProxParams &p, const OffsetView &o, const LeafMap &m, uint32_t n0, const uint32_t *pos,
                        uint8_t *flags, uint32_t *block_counts, hipStream_t s) {
  if (!n0) return;
  const dim3 g(blocks_for(n0)), b(256);
  if (p.n_leaves <= 4) hipLaunchKernelGGL(prox_filter_kernel<4>, g, b, 0, s, p, o, m, n0, pos, flags, block_counts);
  else if (p.n_leaves <= 8) hipLaunchKernelGGL(prox_filter_kernel<8>, g, b, 0, s, p, o, m, n0, pos, flags, block_counts);
  else hipLaunchKernelGGL(prox_filter_kernel<kMaxLists>, g, b, 0, s, p, o, m, n0, pos, flags, block_counts);
}
void launch_prox_slop(const ProxParams &p, const OffsetView &o, const uint32_t *epos, uint32_t len, uint32_t cap,
                      int32_t *slops, hipStream_t s) {
  if (!len) return;
  const dim3 g(blocks_for(len)), b(256);
  if (p.n_leaves <= 4) hipLaunchKernelGGL(prox_slop_kernel<4>, g, b, 0, s, p, o, epos, len, cap, slops);
  else if (p.n_leaves <= 8) hipLaunchKernelGGL(prox_slop_kernel<8>, g, b, 0, s, p, o, epos, len, cap, slops);
  else hipLaunchKernelGGL(prox_slop_kernel<kMaxLists>, g, b, 0, s, p, o, epos, len, cap, slops);
}
void launch_union_flag(const ListView &v, int s, uint8_t *flags, uint32_t *block_counts, hipStream_t st) {
  hipLaunchKernelGGL(union_flag_kernel, dim3(blocks_for(v.len[s])), dim3(256), 0, st, v, s, flags, block_counts);
}
void launch_union_prefix(const uint8_t *flags, uint32_t len, const uint32_t *block_off, const uint32_t *total,
                         uint32_t *prefix, hipStream_t st) {
  hipLaunchKernelGGL(union_prefix_kernel, dim3(blocks_for(len + 1)), dim3(256), 0, st, flags, len, block_off, total, prefix);
}
void launch_union_write(const ListView &v, const LeafMap &m, const UnionView &u, int s, const uint8_t *flags,
                        uint32_t *out_ids, uint32_t *out_freqs, uint32_t cap, hipStream_t st, uint32_t *out_epos) {
  hipLaunchKernelGGL(union_write_kernel, dim3(blocks_for(v.len[s])), dim3(256), 0, st, v, m, u, s, flags, out_ids,
                     out_freqs, cap, out_epos);
}
void launch_not_range(const uint32_t *child, uint32_t child_len, uint32_t max_doc, uint32_t *out_ids,
                      uint32_t *out_freqs, uint32_t cap, hipStream_t st) {
  if (!max_doc) return;
  hipLaunchKernelGGL(not_range_kernel, dim3(blocks_for(max_doc)), dim3(256), 0, st, child, child_len, max_doc, out_ids,
                     out_freqs, cap);
}
void launch_count_below(const uint32_t *list, uint32_t len, uint64_t x, uint32_t *out, hipStream_t st) {
  hipLaunchKernelGGL(count_below_kernel, dim3(1), dim3(1), 0, st, list, len, x, out);
}
void launch_not_universe_flag(const uint32_t *universe, uint32_t n_u, const uint32_t *child, uint32_t child_len,
                              long long child_shift, uint32_t max_doc, uint8_t *flags, uint32_t *block_counts,
                              hipStream_t st) {
  hipLaunchKernelGGL(not_universe_flag_kernel, dim3(blocks_for(n_u)), dim3(256), 0, st, universe, n_u, child, child_len,
                     child_shift, max_doc, flags, block_counts);
}
void launch_not_universe_write(const uint32_t *universe, uint32_t n_u, const uint8_t *flags, const uint32_t *block_off,
                               uint32_t *out_ids, uint32_t *out_freqs, uint32_t cap, hipStream_t st) {
  hipLaunchKernelGGL(not_universe_write_kernel, dim3(blocks_for(n_u)), dim3(256), 0, st, universe, n_u, flags, block_off,
                     out_ids, out_freqs, cap);
}
void launch_score(const ScoreParams &p, const uint32_t *ids, const uint32_t *freqs, uint32_t len, uint32_t cap,
                  const uint32_t *doc_len, const float *doc_score, const uint32_t *max_freq, uint32_t table_n,
                  double *scores, uint64_t *keys, hipStream_t s, uint32_t *keys32) {
  if (!len) return;
  if (p.n_nodes > 0)
    hipLaunchKernelGGL(score_kernel<true>, dim3(blocks_for(len)), dim3(256), 0, s, p, ids, freqs, len, cap, doc_len,
                       doc_score, max_freq, table_n, scores, keys, keys32);
  else
    hipLaunchKernelGGL(score_kernel<false>, dim3(blocks_for(len)), dim3(256), 0, s, p, ids, freqs, len, cap, doc_len,
                       doc_score, max_freq, table_n, scores, keys, keys32);
}
void launch_fetch_cand64(const void *cand, const uint32_t *count, uint32_t cap, const uint64_t *keys64,
                         const uint32_t *ids, uint32_t *out_rows, uint64_t *out_keys, uint32_t *out_ids, uint32_t *out_n,
                         hipStream_t s) {
  hipLaunchKernelGGL(fetch_cand64_kernel, dim3(8), dim3(256), 0, s, (const uint2 *)cand, count, cap, keys64, ids,
                     out_rows, out_keys, out_ids, out_n);
}
// ---- per-hit term records for the iterator seam (query_iterators.c) ----------------------------------------------------
// What the term's own reader would have yielded for hit first+i: out[0] entry index in the term's posting list
// (0xFFFFFFFF: the list does not hold the document -- a union child that did not match), out[1..4] field mask (128 bits),
// out[5] / out[6] position (into the list's bytes) and length of the record's offsets blob; planes of `count` words.
// The entry index comes from the hit list's own column when it has one, else from a binary search of the hit's doc id
// (hit_ids + shift = the id in the list's frame).
__global__ __launch_bounds__(256) void hit_records_kernel(const uint32_t *__restrict__ hit_ids, uint32_t first, uint32_t count,
                                                          const uint32_t *__restrict__ hit_epos,
                                                          const uint32_t *__restrict__ list_ids, uint32_t list_len,
                                                          long long shift, const uint32_t *__restrict__ masks,
                                                          const uint32_t *__restrict__ wmasks,
                                                          const uint32_t *__restrict__ off_pos,
                                                          const uint32_t *__restrict__ off_len, uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  uint32_t e = 0xFFFFFFFFu;
  if (hit_epos) {
    e = hit_epos[first + i];
  } else {
    const long long want = (long long)hit_ids[first + i] + shift;
    if (want >= 0 && want <= 0xFFFFFFFFll) {
      uint32_t lo = 0, hi = list_len;
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((long long)list_ids[mid] < want) lo = mid + 1;
        else hi = mid;
      }
      if (lo < list_len && (long long)list_ids[lo] == want) e = lo;
    }
  }
  const bool have = e != 0xFFFFFFFFu && e < list_len;
  out[i] = have ? e : 0xFFFFFFFFu;
  uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  if (have) {
    if (wmasks) {
      m0 = wmasks[4 * (size_t)e], m1 = wmasks[4 * (size_t)e + 1], m2 = wmasks[4 * (size_t)e + 2], m3 = wmasks[4 * (size_t)e + 3];
    } else if (masks) {
      m0 = masks[e];
    }
  }
  out[(size_t)count + i] = m0;
  out[2 * (size_t)count + i] = m1;
  out[3 * (size_t)count + i] = m2;
  out[4 * (size_t)count + i] = m3;
  out[5 * (size_t)count + i] = have && off_pos ? off_pos[e] : 0;
  out[6 * (size_t)count + i] = have && off_len ? off_len[e] : 0;
}

void launch_hit_records(const uint32_t *hit_ids, uint32_t first, uint32_t count, const uint32_t *hit_epos,
                        const uint32_t *list_ids, uint32_t list_len, long long shift, const uint32_t *masks,
                        const uint32_t *wmasks, const uint32_t *off_pos, const uint32_t *off_len, uint32_t *out, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(hit_records_kernel, dim3(blocks_for(count)), dim3(256), 0, s, hit_ids, first, count, hit_epos, list_ids,
                     list_len, shift, masks, wmasks, off_pos, off_len, out);
}
void launch_gather_u32_counted(const uint32_t *src, uint32_t src_len, const uint32_t *idx, const uint32_t *count,
                               uint32_t cap, uint32_t *out, hipStream_t s) {
  if (!cap) return;
  hipLaunchKernelGGL(gather_u32_counted_kernel, dim3(blocks_for(cap)), dim3(256), 0, s, src, src_len, idx, count, cap, out);
}
void launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t *out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(gather_u32_kernel, dim3(blocks_for(n)), dim3(256), 0, s, src, idx, n, out);
}
void launch_score_max_normalize(double *scores, uint64_t *keys, uint32_t len, uint64_t *max_key_zeroed, hipStream_t s) {
  if (!len) return;
  const uint32_t need = blocks_for(len), grid = need < 1024 ? need : 1024;
  hipLaunchKernelGGL(score_max_kernel, dim3(grid), dim3(256), 0, s, scores, len, (unsigned long long *)max_key_zeroed);
  hipLaunchKernelGGL(score_normalize_kernel, dim3(need), dim3(256), 0, s, scores, keys, len,
                     (const unsigned long long *)max_key_zeroed);
}
void launch_labels_to_rows(const uint32_t *ids, uint32_t n, uint64_t ids_base, const LabelRows &L, uint32_t *rows, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(labels_to_rows_kernel, dim3(blocks_for(n)), dim3(256), 0, s, ids, n, ids_base, L, rows);
}
void launch_labels_to_cand(const uint32_t *ids, uint32_t n, uint64_t ids_base, const LabelRows &L, uint32_t *rows_out, void *cand,
                           uint32_t *count, uint32_t cap, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(labels_to_cand_kernel, dim3(blocks_for(n)), dim3(256), 0, s, ids, n, ids_base, L, rows_out, (uint2 *)cand, count,
                     cap);
}
uint32_t knn_topk_max_k() { return kKnnTopkMaxK; }
size_t knn_topk_scratch_bytes() { return (size_t)kKnnTopkBlocks * kKnnTopkMaxK * sizeof(uint64_t); }
void launch_knn_topk(const float *dists, const void *cand, uint32_t *count, uint32_t cap, uint32_t k, const uint32_t *ids,
                     void *part, uint32_t *ticket, uint32_t *out_rows, uint32_t *out_keys, uint32_t *out_ids, uint32_t *out_n,
                     uint32_t *overflow, hipStream_t s) {
  hipLaunchKernelGGL(knn_topk_kernel, dim3(kKnnTopkBlocks), dim3(256), 0, s, dists, (const uint2 *)cand, count, cap, k, ids,
                     (uint64_t *)part, ticket, out_rows, out_keys, out_ids, out_n, overflow);
}
void launch_dist_to_keys(const float *dists, uint32_t n, uint32_t *keys, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(dist_to_keys_kernel, dim3(blocks_for(n)), dim3(256), 0, s, dists, n, keys);
}

}  // namespace rsgpu
