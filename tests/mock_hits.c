/* tests/mock_hits.c -- a CPU stand-in for the device hit lists behind redisearch_amd/csrc/query_iterators.c, so that the
 * HOST logic of Boundary 3 (Read / SkipTo / Rewind, block paging, the rebuilt result trees) runs in the CPU suite.
 * TEST INFRASTRUCTURE ONLY: it implements the dozen RSGPU_* entry points the iterator library calls (include/rsgpu_search.h)
 * over arrays the test prepared -- typically from the CPU oracle's intersection / union of the same lists -- and is linked
 * with query_iterators.c into tests/_build/libiter_mock.so by tests/test_iterator_host_cpu.py.  The product library
 * (librsgpu_iterators.so) links the real engine instead. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rsgpu_search.h"

#define MAXL 32
struct RSGPU_Postings {
  int codec;
  size_t n_entries;
  const uint8_t *bytes;
  size_t n_bytes;
};
struct RSGPU_Hits {
  size_t len;
  int n_leaves, is_union;
  int order[MAXL];                 /* child slot -> caller's list index */
  const uint64_t *ids;
  const uint32_t *entry[MAXL], *freq[MAXL], *olen[MAXL]; /* by child slot, [len] each */
  const uint64_t *mlo[MAXL], *mhi[MAXL], *opos[MAXL];
  int n_groups;
  int group_first[MAXL + 1], group_op[MAXL];
  double group_weight[MAXL];
  int freed;
};

static struct RSGPU_Hits *g_next;
static int g_reads, g_record_reads, g_byte_reads;
__attribute__((visibility("default"))) void mock_set_next_hits(struct RSGPU_Hits *h) { g_next = h; }
__attribute__((visibility("default"))) void mock_counters(int *reads, int *record_reads, int *byte_reads) {
  *reads = g_reads, *record_reads = g_record_reads, *byte_reads = g_byte_reads;
  g_reads = g_record_reads = g_byte_reads = 0;
}
static struct RSGPU_Hits *take(void) {
  struct RSGPU_Hits *h = g_next;
  g_next = NULL;
  return h;
}
#define API __attribute__((visibility("default")))
API const char *RSGPU_LastError(void) { return "mock"; }
API RSGPU_Hits *RSGPU_IntersectEx(RSGPU_Postings *const *l, size_t n, long s, int o) { (void)l, (void)n, (void)s, (void)o; return take(); }
API RSGPU_Hits *RSGPU_Union(RSGPU_Postings *const *l, size_t n) { (void)l, (void)n; return take(); }
API RSGPU_Hits *RSGPU_Not(RSGPU_Postings *c, RSGPU_Postings *u, uint64_t m) { (void)c, (void)u, (void)m; return take(); }
API RSGPU_Hits *RSGPU_EvalTree(const RSGPU_TreeQuery *q) { (void)q; return take(); }
API void RSGPU_Hits_Free(RSGPU_Hits *h) { if (h) h->freed++; }
API size_t RSGPU_Hits_Len(const RSGPU_Hits *h) { return h ? h->len : 0; }
API int RSGPU_Hits_IsUnion(const RSGPU_Hits *h) { return h && h->is_union; }
API size_t RSGPU_Hits_NumLeaves(const RSGPU_Hits *h) { return h ? (size_t)h->n_leaves : 0; }
API int RSGPU_Hits_LeafOrder(const RSGPU_Hits *h, int *out) {
  for (int i = 0; i < h->n_leaves; i++) out[i] = h->order[i];
  return h->n_leaves;
}
API int RSGPU_Hits_Tree(const RSGPU_Hits *h, int *root_is_union, int *gf, int *gop, double *gw) {
  if (root_is_union) *root_is_union = h->is_union;
  for (int g = 0; g < h->n_groups; g++) {
    if (gf) gf[g] = h->group_first[g];
    if (gop) gop[g] = h->group_op[g];
    if (gw) gw[g] = h->group_weight[g];
  }
  if (gf) gf[h->n_groups] = h->group_first[h->n_groups];
  return h->n_groups;
}
/* the same shape as a post-order node array (what query_iterators.c builds `current` from): every group's leaves, the
 * group's aggregate when it is one, the root last */
API int RSGPU_Hits_TreeNodes(const RSGPU_Hits *h, int *op, int *leaf, int *n_children, double *weight) {
  int n = 0;
  const int ng = h->n_groups ? h->n_groups : h->n_leaves;
  for (int g = 0; g < ng; g++) {
    const int a = h->n_groups ? h->group_first[g] : g, b = h->n_groups ? h->group_first[g + 1] : g + 1;
    const int gop = h->n_groups ? h->group_op[g] : 0;
    for (int l = a; l < b; l++, n++) {
      if (op) op[n] = 0;
      if (leaf) leaf[n] = l;
      if (n_children) n_children[n] = 0;
      if (weight) weight[n] = 1.0;
    }
    if (gop != 0) {
      if (op) op[n] = gop;
      if (leaf) leaf[n] = -1;
      if (n_children) n_children[n] = b - a;
      if (weight) weight[n] = h->group_weight[g];
      n++;
    }
  }
  if (op) op[n] = h->is_union ? 1 : 2;
  if (leaf) leaf[n] = -1;
  if (n_children) n_children[n] = ng;
  if (weight) weight[n] = 1.0;
  return n + 1;
}
API RSGPU_Hits *RSGPU_EvalTreeNodes(const RSGPU_TreeNode *nodes, size_t n, RSGPU_Postings *const *l, size_t nl) {
  (void)nodes, (void)n, (void)l, (void)nl;
  return take();
}
API long RSGPU_Hits_ReadRange(const RSGPU_Hits *h, size_t first, size_t count, uint64_t *ids) {
  g_reads++;
  if (first >= h->len) return 0;
  if (count > h->len - first) count = h->len - first;
  memcpy(ids, h->ids + first, count * sizeof *ids);
  return (long)count;
}
API long RSGPU_Hits_ReadRecords(const RSGPU_Hits *h, size_t list, size_t first, size_t count, uint32_t *entry, uint32_t *freqs,
                                uint64_t *mlo, uint64_t *mhi, uint64_t *opos, uint32_t *olen) {
  g_record_reads++;
  int slot = -1;
  for (int s = 0; s < h->n_leaves; s++)
    if ((size_t)h->order[s] == list) slot = s;
  if (slot < 0) return -1;
  if (first >= h->len) return 0;
  if (count > h->len - first) count = h->len - first;
  memcpy(entry, h->entry[slot] + first, count * 4);
  memcpy(freqs, h->freq[slot] + first, count * 4);
  memcpy(mlo, h->mlo[slot] + first, count * 8);
  memcpy(mhi, h->mhi[slot] + first, count * 8);
  memcpy(opos, h->opos[slot] + first, count * 8);
  memcpy(olen, h->olen[slot] + first, count * 4);
  return (long)count;
}
API int RSGPU_Postings_Codec(const RSGPU_Postings *p) { return p ? p->codec : -1; }
API size_t RSGPU_Postings_NumEntries(const RSGPU_Postings *p) { return p ? p->n_entries : 0; }
API int RSGPU_Postings_ReadBytes(const RSGPU_Postings *p, size_t pos, size_t len, uint8_t *out) {
  g_byte_reads++;
  if (pos > p->n_bytes || len > p->n_bytes - pos) return -1;
  memcpy(out, p->bytes + pos, len);
  return 0;
}
