"""CPU: the register / LDS / scratch budgets the design relies on, read from the built gfx950 code objects
(scripts/kernel_resources.py; no GPU).  A spill or a lost occupancy step does not fail any parity test -- it only makes a
kernel slower -- so the budgets are asserted here:
  * no kernel spills registers, except the one experiment variant behind a knob;
  * scratch memory only where per-thread cursor arrays are the design (the proximity kernels);
  * the query-stationary MFMA pass keeps two wavefronts per SIMD in its default shape (8 waves x 32 queries) and fits its
    LDS ring into the CU's 160 KiB; the tiled GEMM's default ring variant likewise;
  * the single-query scan (scan_kernel<TYPE, METRIC, G, ITERS, U, ...>) never needs more than the 512 registers one wave
    can have; only the deepest unrolling (U = 8 rows in flight per row group) may drop to one wave per SIMD -- its loads in
    flight are what hides the HBM latency then -- every other variant keeps two or more."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
KR = importlib.util.module_from_spec(spec)
spec.loader.exec_module(KR)

LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def kernels():
    if not os.path.isdir(KR.OBJDIR) or not any(f.endswith(".hip.o") for f in os.listdir(KR.OBJDIR)):
        pytest.skip("objects not built (python -m redisearch_amd.build)")
    ks = KR.all_kernels()
    assert len(ks) > 1000                 # every .hip file contributed (the scan alone has ~1600 instantiations)
    return ks


def test_no_spills_on_any_default_path(kernels):
    # (the general hybrid tile kernel parks a dozen and a half scalars -- kernel-argument pointers of its eight lists -- in
    # the lanes of a vector register: v_writelane / v_readlane outside its loops, no memory traffic; VECTOR spills it has none)
    # (decode_dense_kernel: its dense path holds none; the scalars are parked where a group that is NOT dense enters the chain parsers
    # of both lists -- v_writelane at that entry, outside every loop of the dense path)
    spilling = [k["name"] for k in kernels if k["vgpr_spill"] or (k["sgpr_spill"] and not k["name"].startswith(("hybrid_tree_tile_kernel<", "decode_dense_kernel")))]
    # gemm_dma = 2 (24 KiB stages, two workgroups per CU) in filter mode: an A/B knob, never the default (kernels.hpp)
    assert all(re.match(r"gemm_topk_ring_kernel<\d+, 4, 3, 4, 1>", n) for n in spilling), spilling


def test_scratch_only_for_the_proximity_cursors(kernels):
    # (and the deep-tree scorer: one accumulator per nesting level, indexed dynamically -- score_kernel<true> only, the
    # flat / two-level scorer score_kernel<false> must stay scratch-free)
    scratch = {re.sub(r"<.*", "", k["name"]) if not k["name"].startswith("score_kernel") else k["name"]
               for k in kernels if k["scratch"] and not k["vgpr_spill"]}
    # (... and the general hybrid tile kernel, which runs the proximity functions on its compacted candidates)
    assert scratch <= {"prox_filter_kernel", "prox_slop_kernel", "score_kernel<true>", "hybrid_tree_tile_kernel"}, scratch


def test_lds_fits_the_cu(kernels):
    assert max(k["lds"] for k in kernels) <= LDS_PER_CU
    for k in kernels:
        assert k["vgpr"] <= 512, k["name"]


def test_query_stationary_pass_budget(kernels):
    qs = [k for k in kernels if k["name"].startswith("gemm_qs_kernel<")]
    assert len(qs) >= 20
    for k in qs:
        qb = int(re.match(r"gemm_qs_kernel<\d+, \d+, \d+, (\d+)", k["name"]).group(1))
        assert k["wg"] == 512 // qb
        if qb == 1:                       # the default: 8 waves per workgroup = two per SIMD
            assert k["vgpr"] <= 256, (k["name"], k["vgpr"])
        l2 = k["name"].rstrip(">").endswith("true")
        ns = int(re.match(r"gemm_qs_kernel<\d+, \d+, (\d+)", k["name"]).group(1))
        assert k["lds"] <= 147456 + (ns * 256 if l2 else 0)   # the ring (~144 KiB) + the L2 form's 64 half norms per slot


def test_fp32_source_query_stationary_pass_budget(kernels):
    """gemm_qs_f32_kernel<KS, KH, NS, QB, L2> (round 4: fp32 tiles, bf16 in flight): no spills -- a spilled query fragment is
    reloaded through the counted vmcnt queue of the LDS-DMA ring every tile -- two waves per SIMD in the eight-wave shape,
    one 512-register wave in the four-wave shape, the K-part ring inside the CU's LDS."""
    qs = [k for k in kernels if k["name"].startswith("gemm_qs_f32_kernel<")]
    f32 = [k for k in qs if k["name"].rstrip(">").endswith(", 0")]
    h8 = [k for k in qs if k["name"].rstrip(">").endswith(", 1")]      # round 6: SRC_H8, fp16 rows quantised in every wave
    assert len(f32) == 20                 # five row widths x two shapes x {IP / cosine, L2}
    assert len(h8) >= 10                  # five row widths x two shapes (+ the two A/B ring shapes at dim 768)
    for k in qs:
        ks, kh, ns, qb = (int(x) for x in re.match(r"gemm_qs_f32_kernel<(\d+), (\d+), (\d+), (\d+)", k["name"]).groups())
        assert not k["vgpr_spill"] and not k["scratch"], k["name"]
        assert k["wg"] == 512 // qb and k["vgpr"] <= (256 if qb == 1 else 512), (k["name"], k["vgpr"])
        assert k["lds"] <= ns * kh * 2048 + ns * 256 and k["lds"] <= LDS_PER_CU, (k["name"], k["lds"])


def test_tiled_gemm_default_variant_budget(kernels):
    ring = [k for k in kernels if re.match(r"gemm_topk_ring_kernel<\d+, 8, 3, 2, \d>", k["name"])]
    assert len(ring) == 4                 # f16 / bf16 x all-keys / filter
    for k in ring:
        assert k["vgpr"] <= 256 and not k["scratch"], k


def test_single_query_scan_budget(kernels):
    scan = [k for k in kernels if k["name"].startswith("scan_kernel<")]
    assert len(scan) > 1000
    for k in scan:
        u = int(re.match(r"scan_kernel<\d+, \d+, \d+, \d+, (\d+)", k["name"]).group(1))
        assert k["vgpr"] <= (512 if u >= 8 else 256), (k["name"], k["vgpr"])


def test_multi_query_scan_budget(kernels):
    """scan_mq_kernel<TYPE, METRIC, G, ITERS, U, B, EXACT>: eight queries' chunks live in registers (96 of them at 768
    fp32; 192 in the sixteen-query kernel); the kernel must keep two wavefronts per SIMD (<= 256 registers) without spilling
    -- a spill here costs bandwidth, not parity, so only this test would notice."""
    mq = [k for k in kernels if k["name"].startswith("scan_mq_kernel<")]
    assert len(mq) >= 100
    for k in mq:
        assert k["vgpr"] <= 256 and not k["vgpr_spill"] and not k["scratch"], (k["name"], k["vgpr"], k["vgpr_spill"])
    head = [k for k in mq if k["name"].startswith("scan_mq_kernel<0, 1, 64, 3, 4, 8, true")]
    assert len(head) == 1 and head[0]["vgpr"] <= 224, head     # the 10 M x 768 fp32 cosine pass
    assert len([k for k in mq if k["name"].startswith("scan_mq_kernel<0, 1, 64, 3, 2, 16, true")]) == 1   # ... of sixteen


def test_hybrid_two_launch_kernels_budget(kernels):
    """hybrid_tile_kernel keeps seven (six for fp16 / bf16 L2) workgroups per CU -- its phases are dependent memory round trips,
    residency is what hides them: at most 80 registers (six waves per SIMD), no scratch (a select between two structs put the
    first version's keys into scratch memory), static LDS small next to the dynamic pool (16-20 KiB + the query).  The reduce
    kernel is one workgroup of 1 024: at most 128 registers (four waves per SIMD), its survivor lists (4 096 entries since round 6)
    + the list of passing tiles inside 64 KiB.  Round 6: the shared-grid forms (hybrid_tile_batch_kernel / hybrid_reduce_batch_kernel)
    keep the same budgets, and NEITHER tile kernel parks scalars in vector lanes (the first shared-body version of the single-query
    kernel spilled 88: v_readlane in the scorer, 3 us of 45)."""
    for family in ("hybrid_tile_kernel<", "hybrid_tile_batch_kernel<"):
        tiles = [k for k in kernels if k["name"].startswith(family)]
        assert len(tiles) == 6                # FLOAT32 / FLOAT16 / BFLOAT16 x L2 / IP
        for k in tiles:
            assert k["vgpr"] <= 80 and not k["scratch"] and k["lds"] <= 4096 and k["wg"] == 256 and not k["sgpr_spill"], k
    for family in ("hybrid_reduce_kernel", "hybrid_reduce_batch_kernel"):
        red = [k for k in kernels if k["name"].startswith(family)]
        assert len(red) == 1
        assert red[0]["vgpr"] <= 128 and not red[0]["scratch"] and red[0]["lds"] <= 65536 and red[0]["wg"] == 1024, red[0]


def test_general_hybrid_tile_kernel_budget(kernels):
    """hybrid_tree_tile_kernel (round 4): up to eight lists' match positions per driver live in registers through the probe --
    at most 128 registers (four waves per SIMD; the LDS pool of (lists + 1) KiB-words bounds residency before that), scratch only
    for the proximity cursors (ProxCtx<8>: 240 bytes per lane, touched by the lanes that hold a candidate when a window or a
    slop-dependent scorer asks for the term offsets), no vector spills; the instantiations for queries that read no offsets carry
    no cursors and no scratch.  The pack kernel is a copy."""
    tiles = [k for k in kernels if k["name"].startswith("hybrid_tree_tile_kernel<")]
    # round 5: six type / metric pairs x {ML = 4 | 8 with the proximity cursors, ML = 8 without, ML = 8 without + nested trees};
    # round 6: + nested trees WITH the cursors (the root's window / the per-hit slop through nested children)
    assert len(tiles) == 30
    for k in tiles:
        assert k["vgpr"] <= 128 and not k["vgpr_spill"] and k["scratch"] <= 512 and k["lds"] <= 4096 and k["wg"] == 256, k
    # the ML = 4 instantiation (queries of up to four lists) exists for its register budget: six wavefronts per SIMD where the
    # element type allows (<= 80 VGPRs), half the cursor scratch
    small = [k for k in tiles if re.search(r", 4, false, true>$", k["name"])]
    assert len(small) == 6 and all(k["scratch"] <= 128 for k in small) and sum(k["vgpr"] <= 80 for k in small) >= 3, small
    # the queries that read no term offsets (PROX = false; the nested-tree form DEEP among them): no scratch at all
    plain = [k for k in tiles if k["name"].endswith(", false>")]
    assert len(plain) == 12 and all(k["scratch"] == 0 and k["vgpr"] <= 85 for k in plain), plain
    deep = [k for k in plain if k["name"].endswith(", true, false>")]
    assert len(deep) == 6
    pack = [k for k in kernels if k["name"].startswith("hybrid_hits_pack_kernel")]
    assert len(pack) == 1 and pack[0]["vgpr"] <= 32 and not pack[0]["scratch"], pack


def test_register_staged_int8_pass_budget(kernels):
    """gemm_qs_h8r_kernel<KS, 1, D, SRC> (round 6; SRC 1: fp16 rows, BASELINE configs[2]'s default pass, two tiles in flight; SRC 2:
    fp32 rows, one 96 KiB tile in flight): eight waves = two per SIMD (<= 256 registers), NO spill and no scratch -- a spilled ring
    register would be copied while its load is in flight (scripts/isa_lint_h8r.py holds the ISA to that) -- and an int8 double
    buffer of at most 48 KiB."""
    qs = [k for k in kernels if k["name"].startswith("gemm_qs_h8r_kernel<")]
    assert len(qs) == 15                  # five row widths x {fp16 rows (two tiles / A-B: one tile in flight), fp32 rows}
    for k in qs:
        ks, qb, d, src = (int(x) for x in re.match(r"gemm_qs_h8r_kernel<(\d+), (\d+), (\d+), (\d+)", k["name"]).groups())
        assert (qb, d, src) in ((1, 2, 1), (1, 1, 1), (1, 1, 2)), k["name"]
        assert not k["vgpr_spill"] and not k["scratch"], k["name"]
        assert k["wg"] == 512 and k["vgpr"] <= 256, (k["name"], k["vgpr"])
        assert k["lds"] <= 49152, (k["name"], k["lds"])
