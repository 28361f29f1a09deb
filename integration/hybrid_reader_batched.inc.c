/*
 * integration/hybrid_reader_batched.inc.c -- the caller-side batching shim of SURVEY.md 8(f)-2, as real code.
 *
 * The reference's ad-hoc brute-force path asks the vector index for ONE distance per child document
 * (computeDistances_RAM, reference src/iterators/hybrid_reader.c:289-335: a VecSimIndex_GetDistanceFrom_Unsafe call
 * inside the child->Read loop).  On a GPU index that is one kernel launch + one synchronisation per candidate.  The
 * seam already has the batched form -- VecSimIndex_AdhocBfCtx_GetExactDistances, which the reference's disk path uses
 * for re-ranking (hybrid_reader.c:262-266) -- so the shim is a drop-in body for computeDistances_RAM that drains the
 * child in chunks and asks for each chunk's distances with ONE call.
 *
 * This file is THIS repository's code.  It is textually included into a copy of the reference's hybrid_reader.c by
 * oracle/make_batched_hybrid_reader.py (which renames the original function and changes nothing else), so it sees the
 * file's static helpers (insertResultToHeap, vecsimTimeoutCallback) exactly as a maintainer's patch would.
 * tests/test_gpu_reference_hybrid_reader.py runs the reference's iterator both ways on the MI355X engine and requires
 * identical results.
 *
 * Same admission order as the original (candidates are visited in ascending doc id; the running upper bound changes
 * after every admission), so the heap ends up identical.  The context normalises the query itself for cosine
 * (as the disk path relies on, hybrid_reader.c:212-214).
 */
#ifndef RSGPU_ADHOC_CHUNK
#define RSGPU_ADHOC_CHUNK 4096
#endif

static VecSimQueryReply_Code computeDistances_RAM(HybridIterator *hr) {
  double upper_bound = INFINITY;
  VecSimQueryReply_Code rc = VecSim_QueryReply_OK;
  RSIndexResult *cur_vec_res = NewMetricResult();
  VecSimAdhocBfCtx *ctx = VecSimIndex_AdhocBfCtx_New(hr->index, hr->query.vector);
  if (!ctx) { /* an index that does not offer the batched seam: the caller keeps the per-id path */
    IndexResult_Free(cur_vec_res);
    return computeDistances_RAM_perId(hr);
  }
  size_t *labels = rm_malloc(RSGPU_ADHOC_CHUNK * sizeof *labels);
  double *dists = rm_malloc(RSGPU_ADHOC_CHUNK * sizeof *dists);
  /* the child's result is only valid until its next Read: keep a copy unless the heap ignores it anyway */
  RSIndexResult **kept = hr->canTrimDeepResults ? NULL : rm_calloc(RSGPU_ADHOC_CHUNK, sizeof *kept);

  VecSimTieredIndex_AcquireSharedLocks(hr->index);
  IteratorStatus child_status = ITERATOR_OK;
  while (child_status != ITERATOR_EOF && rc == VecSim_QueryReply_OK) {
    size_t n = 0;
    while (n < RSGPU_ADHOC_CHUNK && (child_status = hr->child->Read(hr->child)) == ITERATOR_OK) {
      labels[n] = hr->child->lastDocId;
      if (kept) kept[n] = IndexResult_DeepCopy(hr->child->current);
      n++;
    }
    if (child_status == ITERATOR_TIMEOUT || vecsimTimeoutCallback(&hr->timeoutCtx)) rc = VecSim_QueryReply_TimedOut;
    if (rc == VecSim_QueryReply_OK && n)
      VecSimIndex_AdhocBfCtx_GetExactDistances(ctx, labels, dists, n); /* ONE gather launch; NaN = id has no vector */
    for (size_t i = 0; i < n; i++) {
      if (rc == VecSim_QueryReply_OK && !isnan(dists[i]) &&
          (hr->topResults->count < hr->query.k || dists[i] < upper_bound)) {
        cur_vec_res->docId = labels[i];
        IndexResult_SetNumValue(cur_vec_res, dists[i]);
        insertResultToHeap(hr, kept ? kept[i] : hr->child->current, &cur_vec_res, &upper_bound);
      }
      if (kept && kept[i]) {
        IndexResult_Free(kept[i]);
        kept[i] = NULL;
      }
    }
  }
  VecSimTieredIndex_ReleaseSharedLocks(hr->index);

  VecSimIndex_AdhocBfCtx_Free(ctx);
  rm_free(labels);
  rm_free(dists);
  if (kept) rm_free(kept);
  IndexResult_Free(cur_vec_res);
  return rc;
}
