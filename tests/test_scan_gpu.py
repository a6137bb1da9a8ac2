"""GPU parity suite for the exact vector scan, through the C ABI, against the oracle.

Bar: top-k index sets bit-exact (including order and the chunk_id tie-break), similarities /
distances bit-identical to the fp64-then-cast reference arithmetic (north_star allows 1e-5; we
hold 0 ulp and assert it)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _cases
import _oracle
from yams_amd import _lib
from yams_amd._lib import (SCAN_COSINE, SCAN_L2, FLAG_FORCE_EXACT, FLAG_F32_FILTER, FLAG_SPLIT_FILTER,
                           FLAG_RECORD_PATH, FLAG_WIDE_TILE)

pytestmark = pytest.mark.gpu

TOL = 0.0  # similarities must be bit-identical; the north_star tolerance would be 1e-5


def unblock_i8_shadow(t8, d):
    """The int8 shadow as row-major [padded rows][d]: it is stored blocked — [row / 16][slab][16 rows][4 positions]
    [16 B], position p of row r holding chunk p ^ swz(r), swz = (0, 2, 3, 1)[(r >> 2) & 3] (scan_i8_kernel.hip,
    i8_blocked_offset)."""
    import torch
    npad = t8.numel() // d
    blocked = t8.view(npad // 16, d // 64, 16, 4, 16)
    rowmajor = torch.empty((npad // 16, 16, d // 64, 4, 16), dtype=torch.int8, device=t8.device)
    for g, sw in enumerate((0, 2, 3, 1)):
        for p in range(4):
            rowmajor[:, 4 * g:4 * g + 4, :, p ^ sw, :] = blocked[:, :, 4 * g:4 * g + 4, p, :].permute(0, 2, 1, 3)
    return rowmajor.reshape(npad, d)


# YAMS_TEST_I8_FLAGS=1: every int8 shadow these tests build is in the ROTATED layout (test_int8_tier_tests_pass_in_the_rotated_layout
# runs the tier's tests again that way)
_I8_FLAGS = int(os.environ.get("YAMS_TEST_I8_FLAGS", "0"))


def run(acc, corpus, queries, k, thr=-1.0, metric=SCAN_COSINE, flags=0, tie_rank=None, row_base=0,
        shadow=True, mask=None, i8_flags=None):
    """shadow=True: the corpus view carries the bf16 filter shadow, as the plugin's device mirror
    always does (plugin.cpp corpus_append); shadow=False: a bare fp32 view (flat C-ABI callers);
    shadow="i8": only the INT8 shadow (every batch size takes the int8 tier); shadow="both": both
    (batches above 128 queries take the int8 tier, smaller ones the narrow bf16 form).
    mask: optional boolean allow-mask over the rows."""
    corpus = np.ascontiguousarray(corpus, np.float32)
    dc = acc.to_device(corpus) if corpus.size else None
    db = dn = d8 = dm8 = None
    if shadow in (True, "both") and corpus.size and corpus.shape[1] % 4 == 0:
        db, dn = acc.alloc(corpus.size * 2), acc.alloc(corpus.shape[0] * 4)
        acc.build_shadow_device(dc.ptr, corpus.shape[0], corpus.shape[1], db.ptr, dn.ptr)
    if shadow in ("i8", "both") and corpus.size and corpus.shape[1] % 64 == 0 and corpus.shape[1] >= 256:
        d8, dm8 = acc.alloc(_lib.i8_shadow_rows(corpus.shape[0]) * corpus.shape[1]), acc.alloc((corpus.shape[0] + 15) // 16 * 8)
        if i8_flags is None:
            i8_flags = _I8_FLAGS if corpus.shape[1] <= 4096 else 0
        acc.build_shadow_i8_device(dc.ptr, corpus.shape[0], corpus.shape[1], d8.ptr, dm8.ptr, i8_flags=i8_flags)
    dmask, n_allowed = None, 0
    if mask is not None:
        n = corpus.shape[0]
        bits = np.zeros((n + 31) // 32 * 32, np.uint8); bits[:n] = mask
        words = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
        dmask = acc.to_device(words); n_allowed = int(np.count_nonzero(mask))
    dr = di = None
    if tie_rank is not None:
        inv = np.empty_like(tie_rank)
        inv[tie_rank] = np.arange(tie_rank.size, dtype=tie_rank.dtype)
        dr, di = acc.to_device(tie_rank.astype(np.uint32)), acc.to_device(inv.astype(np.uint32))
    view = acc.corpus_view(dc.ptr if dc else None, corpus.shape[0], corpus.shape[1],
                           dr.ptr if dr else None, di.ptr if di else None, row_base,
                           dmask.ptr if dmask else None, n_allowed,
                           rows_bf16_ptr=db.ptr if db else None, rows_nsq_ptr=dn.ptr if dn else None,
                           rows_i8_ptr=d8.ptr if d8 else None, rows_i8_meta_ptr=dm8.ptr if dm8 else None, i8_flags=(i8_flags or 0) if d8 else 0)
    return acc.scan_topk(view, queries, k, thr, metric, flags)


def check(acc, oracle, corpus, queries, k, thr=-1.0, metric=SCAN_COSINE, flags=0, tie_rank=None,
          max_queries=None, expect_path=None, shadow=True, expect_tier=None, i8_flags=None):
    queries = np.atleast_2d(np.ascontiguousarray(queries, np.float32))
    r = run(acc, corpus, queries, k, thr, metric, flags, tie_rank, shadow=shadow, i8_flags=i8_flags)
    if expect_path is not None:
        assert r.diag["path"] == expect_path, r.diag
    if expect_tier is not None:
        assert r.diag["filter_tier"] == expect_tier, r.diag
    tr64 = None if tie_rank is None else tie_rank.astype(np.uint64)
    nq = queries.shape[0]
    n_single = nq if max_queries is None else min(nq, max_queries)
    # EVERY query of a larger batch through the batched oracle drivers (tests/test_oracle.py pins them to the
    # single-query functions); `max_queries` then only bounds how many ALSO go through the single-query functions
    if tie_rank is None and nq >= _oracle.MANY_FROM and n_single < nq:
        many = oracle.scan_cosine_many(corpus, queries, k, thr) if metric == SCAN_COSINE else oracle.scan_l2_many(corpus, queries, k)
        if many is not None:
            for qi in range(nq):
                if metric == SCAN_COSINE:
                    cnt = int(many[2][qi]); rows, sims, dist = many[0][qi, :cnt], many[1][qi, :cnt], None
                else:
                    cnt = int(many[3][qi])
                    keep = ~(many[2][qi, :cnt] < np.float32(thr))         # vec0: the k nearest, THEN the cosine threshold
                    rows, dist, sims = many[0][qi, :cnt][keep], many[1][qi, :cnt][keep], many[2][qi, :cnt][keep]
                cnt = int(r.counts[qi])
                assert cnt == len(rows), (qi, cnt, len(rows), r.diag)
                assert np.array_equal(r.rows[qi, :cnt], rows), (qi, r.rows[qi, :cnt][:10], rows[:10], r.diag)
                assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), qi
                if dist is not None:
                    assert np.array_equal(r.dist[qi, :cnt].view(np.uint32), dist.view(np.uint32)), qi
                assert (r.rows[qi, cnt:] == -1).all()
            n_single = min(n_single, 2)
    for qi in range(n_single):
        if metric == SCAN_COSINE:
            rows, sims, _, _ = oracle.scan_cosine(corpus, queries[qi], k, thr, tr64)
            dist = None
        else:
            rows, dist, sims = oracle.scan_l2(corpus, queries[qi], k, thr, tr64)
        cnt = int(r.counts[qi])
        assert cnt == len(rows), (qi, cnt, len(rows), r.diag)
        assert np.array_equal(r.rows[qi, :cnt], rows), (qi, r.rows[qi, :cnt][:10], rows[:10], r.diag)
        assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), qi
        if dist is not None:
            assert np.array_equal(r.dist[qi, :cnt].view(np.uint32), dist.view(np.uint32)), qi
        assert (r.rows[qi, cnt:] == -1).all()
    assert r.diag["used_exact_scan"] == 1 and r.diag["rows_visited"] == nq * corpus.shape[0]
    assert r.diag["exact_distance_evaluations"] == nq * corpus.shape[0]
    return r


# ---- the reference's own known-answer tests, replayed through the C ABI --------------------------
def test_reference_exact_scan_contract(acc, oracle):
    # vector_smoke_catch2_test.cpp:188-225 (reference): 6 rows {1,i,0,0}, query {1,0,0,0}, k=3
    c = np.array([[1.0, i, 0, 0] for i in range(6)], np.float32)
    r = check(acc, oracle, c, [1, 0, 0, 0], 3)
    assert r.rows[0, 0] == 0 and r.counts[0] == 3
    assert r.diag["rows_visited"] == 6 and r.diag["exact_distance_evaluations"] == 6
    assert r.diag["returned_rows"] == 3


def test_reference_rejects_invalid_queries(acc):
    # vector_smoke_catch2_test.cpp:227-261 (reference): zero norm / NaN -> InvalidArgument
    c = np.array([[1.0, 0, 0, 0]], np.float32)
    for bad in ([0, 0, 0, 0], [1, np.nan, 0, 0], [np.inf, 0, 0, 0], [1e-6, 0, 0, 0]):
        with pytest.raises(_lib.AccelError) as e:
            run(acc, c, np.array(bad, np.float32), 1)
        assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    # a batch fails as a whole (sqlite_vec_backend.cpp:1635-1647)
    with pytest.raises(_lib.AccelError):
        run(acc, c, np.array([[1, 0, 0, 0], [0, 0, 0, 0]], np.float32), 1)
    # k == 0 returns empty BEFORE validation (:4123-4126)
    r = run(acc, c, np.array([0, 0, 0, 0], np.float32), 0)
    assert r.counts[0] == 0


def test_large_dimensions_up_to_the_limit_and_a_clean_refusal_above(acc, oracle):
    """The fp64 re-score stages the query and its candidate rows in LDS: every dimension up to
    YAMS_SCAN_MAX_DIM (8192) must launch on every path (filter + re-score, widened, exhaustive), a larger
    one is refused up front with YAMS_ERR_UNSUPPORTED instead of failing a launch somewhere inside."""
    rng = np.random.default_rng(5)
    for d in (4096, 8192):
        corpus = oracle.synth_rows(50, 0, 5000, d)
        q = oracle.synth_rows(50, 1 << 40, 2, d)
        corpus[10:4000] = (corpus[7] * rng.uniform(0.5, 2.0, (3990, 1))).astype(np.float32)   # thousands of exact ties: widen + fallback
        check(acc, oracle, corpus, np.stack([q[0], corpus[7]]), 50, max_queries=2)
        check(acc, oracle, corpus[:600], q, 700, flags=FLAG_FORCE_EXACT, expect_path=1)
    with pytest.raises(_lib.AccelError) as e:
        run(acc, np.ones((8, 8196), np.float32), np.ones(8196, np.float32), 1, shadow=False)
    assert e.value.status == _lib.YAMS_ERR_UNSUPPORTED


def test_reference_large_finite_scores(acc, oracle):
    # vector_smoke_catch2_test.cpp:263-302 (reference): +-FLT_MAX/4 self match stays finite, > 0.999
    L = np.float32(np.finfo(np.float32).max / 4)
    e = np.array([[L, -L, L, -L]], np.float32)
    r = check(acc, oracle, e, e[0], 1)
    assert np.isfinite(r.scores[0, 0]) and r.scores[0, 0] > 0.999
    # the same rows inside a corpus large enough for the MFMA filter (fp32 overflows there and the
    # row must be routed to the fp64 re-score)
    corpus = oracle.synth_rows(1, 0, 6000, 4)
    corpus[100] = e[0]; corpus[5000] = -e[0]
    r = check(acc, oracle, corpus, np.stack([e[0], corpus[7]]), 5, expect_path=0)
    assert r.rows[0, 0] == 100


def test_reference_tie_break_by_chunk_id(acc, oracle):
    # vector_smoke_catch2_test.cpp:304-353 (reference)
    for ids in (["tie_c", "tie_a", "tie_b"], ["tie_b", "tie_a", "tie_c"]):
        rank, _ = _cases.string_ranks(ids)
        c = np.tile(np.array([1, 0, 0, 0], np.float32), (3, 1))
        r = run(acc, c, np.array([1, 0, 0, 0], np.float32), 2, -1.0, tie_rank=rank)
        assert [ids[i] for i in r.rows[0, :2]] == ["tie_a", "tie_b"]


def test_reference_fixture_cases(acc, oracle):
    # sqlite_vec_backend_comprehensive_catch2_test.cpp:815-844, 1152-1181, 1451-1478 (reference)
    corpus = np.stack([_cases.fixture_embedding(64, s + 1) for s in range(10)])
    r = check(acc, oracle, corpus, _cases.fixture_embedding(64, 1), 5, 0.0)
    assert r.rows[0, 0] == 0 and r.counts[0] == 5
    r = check(acc, oracle, corpus[:3], corpus[0], 100, 0.0)          # k > corpus
    assert 1 <= r.counts[0] <= 3
    assert check(acc, oracle, corpus, corpus[0], 10, 0.99).counts[0] <= 2
    assert check(acc, oracle, corpus, corpus[0], 10, -1.0).counts[0] == 10
    # 64 x 64 self query, exactDistanceEvaluations == corpus (:1068-1112)
    c64 = np.stack([_cases.fixture_embedding(64, s + 1) for s in range(64)])
    r = check(acc, oracle, c64, c64[17], 1)
    assert r.rows[0, 0] == 17 and r.diag["exact_distance_evaluations"] == 64


def test_empty_and_degenerate(acc):
    r = run(acc, np.zeros((0, 8), np.float32), np.ones(8, np.float32), 5)
    assert r.counts[0] == 0 and r.diag["rows_visited"] == 0
    c = np.zeros((10, 8), np.float32)                      # only zero-norm rows: all skipped
    assert run(acc, c, np.ones(8, np.float32), 5).counts[0] == 0
    with pytest.raises(_lib.AccelError):                   # k beyond the supported bound
        run(acc, np.ones((4, 8), np.float32), np.ones(8, np.float32), 5000)


# ---- seeded parity sweeps ---------------------------------------------------------------------
@pytest.mark.parametrize("n,d,nq,k,metric,thr", [
    (1, 4, 1, 1, SCAN_COSINE, -1.0),
    (100, 3, 2, 7, SCAN_COSINE, -1.0),            # dim not a multiple of 4 -> fp64 path
    (3000, 64, 5, 10, SCAN_COSINE, 0.0),
    (3000, 384, 3, 100, SCAN_L2, -1.0),
    (4096, 128, 9, 10, SCAN_COSINE, -1.0),        # few queries on a small corpus: the fused one-launch path
    (4096, 128, 19, 10, SCAN_COSINE, -1.0),       # smallest MFMA-path corpus (more than 16 queries)
    (5001, 100, 4, 20, SCAN_COSINE, -1.0),        # ragged tile + dim % 32 != 0
    (20000, 384, 37, 100, SCAN_COSINE, -1.0),
    (20000, 768, 130, 10, SCAN_L2, -1.0),
    (150000, 384, 130, 100, SCAN_COSINE, -1.0),   # sample + filter passes, 2 query tiles
    (150000, 384, 40, 100, SCAN_COSINE, 0.16),    # threshold cuts the result short
    (150000, 256, 16, 100, SCAN_L2, 0.05),        # vec0: top-k by distance, then cosine threshold
    (60000, 128, 8, 1000, SCAN_COSINE, -1.0),     # large k: split filter
    (60000, 128, 8, 500, SCAN_COSINE, -1.0),      # k = 500: still the single-pass filter (1564 candidates)
    (50000, 96, 5, 300, SCAN_L2, -1.0),           # L2 k = 300: single pass (1928 candidates)
    (33000, 1024, 3, 1, SCAN_COSINE, -1.0),
])
def test_parity_sweep(acc, oracle, n, d, nq, k, metric, thr):
    corpus = oracle.synth_rows(7, 0, n, d)
    queries = oracle.synth_rows(7, 1 << 40, nq, d)
    fused = n <= 16384 and nq <= 16 and d % 32 == 0          # scan_small_kernel: one launch, every row in fp64
    path = 0 if (n >= 4096 and d % 4 == 0 and not fused) else 1
    check(acc, oracle, corpus, queries, k, thr, metric, max_queries=24, expect_path=path)


def test_unnormalised_and_scaled_vectors(acc, oracle):
    rng = np.random.default_rng(31)
    corpus = (rng.standard_normal((30000, 96)) * rng.uniform(1e-3, 1e3, (30000, 1))).astype(np.float32)
    queries = (rng.standard_normal((6, 96)) * 50).astype(np.float32)
    check(acc, oracle, corpus, queries, 25, expect_path=0)
    check(acc, oracle, corpus, queries, 25, metric=SCAN_L2, expect_path=0)


def test_adversarial_rows_on_the_filter_path(acc, oracle):
    """Zero rows, NaN/inf rows, tiny and huge norms, duplicates and scaled copies (exact fp64 ties
    that differ in fp32) inside a corpus that takes the MFMA filter; ties go by a shuffled rank."""
    rng = np.random.default_rng(32)
    n, d = 12000, 64
    corpus = oracle.synth_rows(9, 0, n, d)
    q = oracle.synth_rows(9, 1 << 40, 4, d)
    corpus[10] = 0
    corpus[11, 3] = np.nan
    corpus[12, 5] = np.inf
    corpus[13] *= np.float32(1e-25)        # norm^2 ~ 1e-50 <= 1e-12 -> skipped by the reference
    corpus[14] *= np.float32(3e-7)         # norm^2 ~ 9e-14 <= 1e-12 -> skipped
    corpus[15] *= np.float32(1e18)
    corpus[16] *= np.float32(np.finfo(np.float32).max / 8)
    best = int(oracle.scan_cosine(corpus, q[0], 1)[0][0])
    for j, s in enumerate([2.0, 0.5, 3.0, 1.0, 7.0, 1.0]):   # scaled + exact duplicates of the winner
        corpus[2000 + 17 * j] = corpus[best] * np.float32(s)
    corpus[9000:9040] = corpus[best]                       # 40 exact duplicates (> k ties at rank 1)
    rank = rng.permutation(n).astype(np.uint32)
    for k in (1, 10, 60):
        r = check(acc, oracle, corpus, q, k, tie_rank=rank)
        check(acc, oracle, corpus, q, k, metric=SCAN_L2, tie_rank=rank)
    assert r.diag["path"] == 0


@pytest.mark.parametrize("flags,ties", [(0, 260), (FLAG_SPLIT_FILTER, 120)])
def test_ties_beyond_the_candidate_budget_are_widened(acc, oracle, flags, ties):
    """More identical rows than the first-stage candidate budget (k' = 224 for the single-pass
    filter, 96 for the split filter at k = 50) but fewer than the candidate list holds: stage 1
    cannot prove completeness, stage 2 re-scores the whole list."""
    n, d = 20000, 32
    corpus = oracle.synth_rows(10, 0, n, d)
    q = oracle.synth_rows(10, 1 << 40, 3, d)
    corpus[100:20000:40][:ties] = q[1]                      # `ties` rows tie with similarity 1.0
    rank = np.random.default_rng(33).permutation(n).astype(np.uint32)
    r = check(acc, oracle, corpus, q, 50, flags=flags, tie_rank=rank, expect_path=0)
    assert r.diag["widened_queries"] >= 1 and r.diag["exact_fallback_queries"] == 0
    assert r.diag["escalated_queries"] == 0


def test_crowded_top_escalates_to_the_split_filter(acc, oracle):
    """3000 rows within 0.009 of each other at the top: far more than the single-pass filter
    (bound 2^-7) can separate within its candidate budget, trivially separable for the split
    filter (bound 6e-5).  The unproven queries are re-run through the split filter on the device;
    nothing reaches the exhaustive scan and the other queries are untouched."""
    n, d = 20000, 64
    rng = np.random.default_rng(91)
    corpus = oracle.synth_rows(12, 0, n, d)
    q = oracle.synth_rows(12, 1 << 40, 5, d)
    qu = (q[2] / np.linalg.norm(q[2])).astype(np.float64)
    crowd = rng.choice(n, 3000, replace=False)
    sims = np.linspace(0.990, 0.999, 3000)
    for r_, s_ in zip(crowd, sims):
        v = rng.standard_normal(d)
        v -= (v @ qu) * qu
        v /= np.linalg.norm(v)
        corpus[r_] = (s_ * qu + np.sqrt(1.0 - s_ * s_) * v).astype(np.float32) * np.float32(rng.uniform(0.5, 2.0))
    r = check(acc, oracle, corpus, q, 50, expect_path=0)
    assert r.diag["escalated_queries"] >= 1 and r.diag["exact_fallback_queries"] == 0, r.diag
    r3 = check(acc, oracle, corpus, q, 50, flags=FLAG_SPLIT_FILTER, expect_path=0)
    assert r3.diag["escalated_queries"] == 0 and r3.diag["exact_fallback_queries"] == 0, r3.diag
    check(acc, oracle, corpus, q, 50, metric=SCAN_L2, expect_path=0)


def test_escalation_and_list_overflow_in_one_batch(acc, oracle):
    """One batch holding a list-overflow query at index 0 AND a crowded-top query: the crowded one is
    re-run through the split filter as a nested one-query batch (slot 0 of ITS workspace) before
    the overflowed one takes the exhaustive fp64 pass with the outer call's query norms.  The two
    queries have different norms, so a nested run that shared the outer workspace would score query
    0 with the wrong norm (round-1 advisor finding)."""
    n, d = 20000, 64
    rng = np.random.default_rng(92)
    corpus = oracle.synth_rows(13, 0, n, d)
    q = oracle.synth_rows(13, 1 << 40, 5, d)
    q[0] *= np.float32(3.0)
    corpus[1::4] = q[0] * np.float32(0.5)                   # 5000 exact ties: query 0's list overflows
    qu = (q[2] / np.linalg.norm(q[2])).astype(np.float64)
    pool = np.setdiff1d(np.arange(n), np.arange(1, n, 4))
    crowd = rng.choice(pool, 3000, replace=False)
    for r_, s_ in zip(crowd, np.linspace(0.990, 0.999, 3000)):
        v = rng.standard_normal(d)
        v -= (v @ qu) * qu
        v /= np.linalg.norm(v)
        corpus[r_] = (s_ * qu + np.sqrt(1.0 - s_ * s_) * v).astype(np.float32) * np.float32(rng.uniform(0.5, 2.0))
    rank = np.random.default_rng(34).permutation(n).astype(np.uint32)
    r = check(acc, oracle, corpus, q, 50, tie_rank=rank, expect_path=0)
    assert r.diag["escalated_queries"] >= 1 and r.diag["exact_fallback_queries"] >= 1, r.diag


def test_product_library_ignores_measurement_environment(acc, oracle, monkeypatch):
    """YAMS_ACCEL_BF16_KERNEL / _PASSES select ablation kernels in the measurement build only
    (-DYAMS_ACCEL_MEASURE, scripts/); the product library must not read them."""
    monkeypatch.setenv("YAMS_ACCEL_BF16_KERNEL", "12")
    monkeypatch.setenv("YAMS_ACCEL_BF16_PASSES", "3")
    monkeypatch.setenv("YAMS_ACCEL_CDC_GENERIC", "1")
    corpus = oracle.synth_rows(18, 0, 30000, 128)
    q = oracle.synth_rows(18, 1 << 40, 4, 128)
    r = check(acc, oracle, corpus, q, 20, expect_path=0)
    assert r.diag["escalated_queries"] == 0 and (r.counts == 20).all()


def test_tau_select_with_more_tied_group_maxima_than_its_list_holds(acc, oracle):
    """tau_select_kernel's two-pass form collects the keys at or above a lower bound into a 2048-entry list in LDS;
    when more group maxima than that TIE at the top it must fall through to the radix select.  2.3M rows of which
    every second one is the same vector (36k sampled rows = 2250 sample groups, every one of them holding a copy):
    one query equal to that vector (its threshold is the tie value, its candidate list overflows, the exhaustive
    fp64 path answers), two ordinary queries beside it; tie ranks decide among the 1.15M rows at similarity 1."""
    n, d = 2_300_000, 32
    corpus = oracle.synth_rows(51, 0, n, d)
    q = oracle.synth_rows(51, 1 << 40, 3, d)
    corpus[::2] = q[1]
    rank = np.random.default_rng(51).permutation(n).astype(np.uint32)
    r = check(acc, oracle, corpus, q, 10, tie_rank=rank, expect_path=0)
    assert r.diag["exact_fallback_queries"] >= 1, r.diag


def test_measurement_build_filter_forms_agree_with_the_product_form():
    """The two alternative forms of the resident-query int8 filter kept in the measurement build (128 x 128
    wave tiles with one wave per SIMD; the product's tiles with direct row loads) are not dead code paths: on ragged
    shards, with thresholds and an allow-mask, they give the product form's results bit for bit AND its candidate
    sets (tests/_filter_forms.py, its own process: the measurement library is chosen at import)."""
    import json, subprocess, sys
    from yams_amd import build as _build
    if not os.path.exists(_build.MEASURE_LIB):
        pytest.skip("the measurement build is not in the tree (python -m yams_amd.build --measure)")
    env = dict(os.environ, YAMS_ACCEL_MEASURE_LIB="1")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_filter_forms.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    recs = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(recs) == 5
    for r in recs:
        assert r["tier"] == _lib.TIER_I8, r
        assert r["identical_70"] and r["identical_80"] and r["identical_87"] and r["identical_90"], r
        assert r["candidates"]["70"] == r["candidates"]["2"] == r["candidates"]["80"] == r["candidates"]["87"] == r["candidates"]["90"], r
        assert r["fallback"] == {"2": 0, "70": 0, "80": 0, "87": 0, "90": 0}, r


def test_massive_ties_take_the_exhaustive_fp64_path(acc, oracle):
    """More identical rows than the candidate list can hold: the list overflows, the query is
    scored exhaustively in fp64 on the device — results stay exact, other queries stay fast."""
    n, d = 20000, 32
    corpus = oracle.synth_rows(10, 0, n, d)
    q = oracle.synth_rows(10, 1 << 40, 3, d)
    corpus[::4] = q[1]                                      # 5000 rows tie with similarity 1.0
    rank = np.random.default_rng(33).permutation(n).astype(np.uint32)
    r = check(acc, oracle, corpus, q, 50, tie_rank=rank, expect_path=0)
    assert 1 <= r.diag["exact_fallback_queries"] <= 2


def test_sorted_corpus_defeats_the_sample_but_not_the_result(acc, oracle):
    """Rows ordered by similarity to the query: every tile beats the sampled threshold's
    expectation; the candidate lists may overflow — the answer must not change."""
    n, d = 60000, 64
    corpus = oracle.synth_rows(11, 0, n, d)
    q = oracle.synth_rows(11, 1 << 40, 2, d)
    order = np.argsort(corpus @ q[0])
    check(acc, oracle, corpus[order], q, 100)
    check(acc, oracle, corpus[order[::-1]], q, 100)


@pytest.mark.parametrize("metric", [SCAN_COSINE, SCAN_L2])
def test_all_filters_agree_with_the_oracle(acc, oracle, metric):
    """The single-pass bf16 filter (default), the split-bf16 filter and the exact-f32 filter are
    interchangeable: all are only filters in front of the same fp64 re-score + proof."""
    corpus = oracle.synth_rows(16, 0, 70000, 256)
    q = oracle.synth_rows(16, 1 << 40, 300, 256)          # > 256 queries: two bf16 query tiles
    a = check(acc, oracle, corpus, q, 50, metric=metric, max_queries=12, expect_path=0)
    b = check(acc, oracle, corpus, q, 50, metric=metric, flags=FLAG_F32_FILTER, max_queries=12, expect_path=0)
    c = check(acc, oracle, corpus, q, 50, metric=metric, flags=FLAG_SPLIT_FILTER, max_queries=12, expect_path=0)
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.scores.view(np.uint32), b.scores.view(np.uint32))
    assert np.array_equal(a.rows, c.rows) and np.array_equal(a.scores.view(np.uint32), c.scores.view(np.uint32))
    assert c.diag["exact_fallback_queries"] == 0 and a.diag["escalated_queries"] == 0
    # without the shadow the single-pass filter converts the fp32 rows in its loop
    p = check(acc, oracle, corpus, q, 50, metric=metric, max_queries=12, expect_path=0, shadow=False)
    assert np.array_equal(a.rows, p.rows) and np.array_equal(a.scores.view(np.uint32), p.scores.view(np.uint32))
    # dim 240 (multiple of 16, not of 32) takes the 16-wide-slab form of the single-pass kernel
    check(acc, oracle, corpus[:30000, :240].copy(), q[:40, :240].copy(), 20, metric=metric, max_queries=6, expect_path=0)
    assert a.diag["exact_fallback_queries"] == 0 and b.diag["exact_fallback_queries"] == 0


@pytest.mark.parametrize("metric", [SCAN_COSINE, SCAN_L2])
@pytest.mark.parametrize("nq", [1, 33, 64, 65, 128, 129])
def test_small_batches_take_the_narrow_filter_and_agree(acc, oracle, metric, nq):
    """Batches of <= 64 / <= 128 queries run the narrow (HBM-bound) form of the shadow filter; 129
    falls back to the 256-query tile.  Same result as the oracle and, bit for bit, as the wide form
    (YAMS_SCAN_FLAG_WIDE_TILE); ragged row tail, zero-norm and huge-norm rows, thresholds."""
    n, d, k = 40000 + 77, 192, 30
    corpus = oracle.synth_rows(21, 0, n, d)
    corpus[5] = 0.0
    corpus[n - 1] = 0.0
    corpus[777] *= 1e18          # out-of-range norm: NaN filter score, exact re-score decides
    q = oracle.synth_rows(21, 1 << 40, nq, d)
    thr = 0.05 if metric == SCAN_COSINE else -1.0
    a = check(acc, oracle, corpus, q, k, thr=thr, metric=metric, max_queries=5, expect_path=0)
    b = run(acc, corpus, q, k, thr, metric, flags=FLAG_WIDE_TILE)
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.counts, b.counts)
    assert np.array_equal(a.scores.view(np.uint32), b.scores.view(np.uint32))
    assert a.diag["exact_fallback_queries"] == 0


def test_forced_exact_equals_filter_path(acc, oracle):
    corpus = oracle.synth_rows(12, 0, 50000, 128)
    q = oracle.synth_rows(12, 1 << 40, 6, 128)
    a = run(acc, corpus, q, 40)
    b = run(acc, corpus, q, 40, flags=FLAG_FORCE_EXACT)
    assert a.diag["path"] == 0 and b.diag["path"] == 1
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.scores.view(np.uint32), b.scores.view(np.uint32))


def test_row_base_and_device_entry_point(acc, oracle):
    import torch
    n, d, nq, k = 10000, 64, 3, 5
    corpus = oracle.synth_rows(13, 0, n, d)
    q = oracle.synth_rows(13, 1 << 40, nq, d)
    tc, tq = torch.from_numpy(corpus).cuda(), torch.from_numpy(q).cuda()
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    rws = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
    rk = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    view = acc.corpus_view(tc.data_ptr(), n, d, row_base=1_000_000)
    acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), rws.data_ptr(),
                         cnt.data_ptr(), None, rk.data_ptr())
    for qi in range(nq):
        rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k)
        assert np.array_equal(rws[qi].cpu().numpy(), rows + 1_000_000)
        assert np.array_equal(rk[qi].cpu().numpy().astype(np.int64), rows)


# ---- filtered search: document_hash / candidate_hashes as a row allow-mask -----------------------
def _masked(acc, oracle, corpus, queries, k, allowed, metric=SCAN_COSINE, thr=-1.0, tie_rank=None):
    n, d = corpus.shape
    bits = np.zeros((n + 31) // 32 * 32, np.uint8)
    bits[allowed] = 1
    words = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
    dc, dm = acc.to_device(corpus), acc.to_device(words)
    dr = di = None
    if tie_rank is not None:
        inv = np.empty_like(tie_rank); inv[tie_rank] = np.arange(n, dtype=tie_rank.dtype)
        dr, di = acc.to_device(tie_rank), acc.to_device(inv)
    view = acc.corpus_view(dc.ptr, n, d, dr.ptr if dr else None, di.ptr if di else None, 0, dm.ptr, len(allowed))
    r = acc.scan_topk(view, queries, k, thr, metric)
    if d % 32 == 0 and n >= 4096:
        # the same search over a mirror that carries the bf16 shadow (as the plugin's always does):
        # small batches then take the narrow filter, whose epilogue applies the same mask
        db, dn = acc.alloc(corpus.size * 2), acc.alloc(n * 4)
        acc.build_shadow_device(dc.ptr, n, d, db.ptr, dn.ptr)
        view_s = acc.corpus_view(dc.ptr, n, d, dr.ptr if dr else None, di.ptr if di else None, 0, dm.ptr,
                                 len(allowed), rows_bf16_ptr=db.ptr, rows_nsq_ptr=dn.ptr)
        rs = acc.scan_topk(view_s, queries, k, thr, metric)
        assert np.array_equal(rs.rows, r.rows) and np.array_equal(rs.counts, r.counts)
        assert np.array_equal(rs.scores.view(np.uint32), r.scores.view(np.uint32))
        assert rs.diag["path"] == r.diag["path"] and rs.diag["rows_visited"] == r.diag["rows_visited"]
    sub = corpus[allowed]
    sub_rank = None if tie_rank is None else tie_rank[allowed].astype(np.uint64)
    for qi in range(queries.shape[0]):
        if metric == SCAN_COSINE:
            rows, sims, _, _ = oracle.scan_cosine(sub, queries[qi], k, thr, sub_rank)
        else:
            rows, _, sims = oracle.scan_l2(sub, queries[qi], k, thr, sub_rank)
        cnt = int(r.counts[qi])
        assert cnt == len(rows), (qi, cnt, len(rows), r.diag)
        assert np.array_equal(r.rows[qi, :cnt], np.asarray(allowed)[rows]), qi
        assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32))
    # only the allowed rows are visited / evaluated (vector_smoke_catch2_test.cpp:355-401)
    assert r.diag["rows_visited"] == queries.shape[0] * len(allowed)
    assert r.diag["exact_distance_evaluations"] == queries.shape[0] * len(allowed)
    return r


def test_reference_candidate_mode_scores_only_allowed_documents(acc, oracle):
    c = np.array([[1, 0, 0, 0], [0.8, 0.6, 0, 0], [1, 0, 0, 0]], np.float32)   # allowed, allowed, blocked
    r = _masked(acc, oracle, c, np.array([[1, 0, 0, 0]], np.float32), 4, [0, 1])
    assert list(r.rows[0, :2]) == [0, 1] and r.counts[0] == 2 and r.diag["returned_rows"] == 2


@pytest.mark.parametrize("metric", [SCAN_COSINE, SCAN_L2])
def test_row_mask_dense_and_sparse(acc, oracle, metric):
    rng = np.random.default_rng(51)
    n, d = 60000, 128
    corpus = oracle.synth_rows(17, 0, n, d)
    q = oracle.synth_rows(17, 1 << 40, 5, d)
    rank = rng.permutation(n).astype(np.uint32)
    dense = np.sort(rng.choice(n, 40000, replace=False))          # MFMA filter with masked rows
    r = _masked(acc, oracle, corpus, q, 30, dense, metric, tie_rank=rank)
    assert r.diag["path"] == 0
    sparse = np.sort(rng.choice(n, 700, replace=False))           # gathered fp64 path
    r = _masked(acc, oracle, corpus, q, 30, sparse, metric, tie_rank=rank)
    assert r.diag["path"] == 1
    # the best unmasked row must not leak: mask out each query's global winner
    best = [int(oracle.scan_cosine(corpus, q[i], 1)[0][0]) for i in range(5)]
    keep = np.setdiff1d(np.arange(n), best)
    _masked(acc, oracle, corpus, q, 10, keep, metric)
    _masked(acc, oracle, corpus, q, 10, [], metric)               # empty candidate set -> no results


def test_record_path_metadata_filters(acc, oracle):
    """YAMS_SCAN_FLAG_RECORD_PATH = the reference's metadata-filter path (:4333-4409): the host folds
    the metadata predicate into the allow-mask; rows with norm^2 in [1e-12, 1e-10) are dropped there
    but kept by the fast path.  Dense mask -> MFMA filter, sparse mask -> gathered fp64 path."""
    rng = np.random.default_rng(77)
    n, d = 50000, 64
    corpus = oracle.synth_rows(19, 0, n, d)
    q = oracle.synth_rows(19, 1 << 40, 4, d)
    qu = q / np.linalg.norm(q, axis=1, keepdims=True)
    # rows parallel to the queries with norm^2 = 2.5e-11: cosine 1.0 — winners for the fast path only
    for i in range(4):
        corpus[1000 + i] = (qu[i] * 5e-6).astype(np.float32)
    rank = rng.permutation(n).astype(np.uint32)
    inv = np.empty_like(rank); inv[rank] = np.arange(n, dtype=rank.dtype)
    for allowed in (np.sort(np.concatenate([np.arange(1000, 1004), rng.choice(n, 30000, replace=False)])),
                    np.sort(np.concatenate([np.arange(1000, 1004), rng.choice(n, 500, replace=False)]))):
        allowed = np.unique(allowed)
        bits = np.zeros((n + 31) // 32 * 32, np.uint8); bits[allowed] = 1
        words = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
        dc, dm, dr, di = acc.to_device(corpus), acc.to_device(words), acc.to_device(rank), acc.to_device(inv)
        view = acc.corpus_view(dc.ptr, n, d, dr.ptr, di.ptr, 0, dm.ptr, len(allowed))
        allow8 = bits[:n]
        fast = acc.scan_topk(view, q, 10, -1.0, SCAN_COSINE, 0)
        rec = acc.scan_topk(view, q, 10, -1.0, SCAN_COSINE, FLAG_RECORD_PATH)
        for qi in range(4):
            rows, sims, _ = oracle.scan_cosine_records(corpus, q[qi], 10, -1.0, rank.astype(np.uint64), allow8)
            assert np.array_equal(rec.rows[qi, :rec.counts[qi]], rows), (qi, rec.diag)
            assert np.array_equal(rec.scores[qi, :rec.counts[qi]].view(np.uint32), sims.view(np.uint32))
            assert 1000 + qi not in rec.rows[qi]            # dropped: norm^2 < 1e-10
            assert fast.rows[qi, 0] == 1000 + qi            # kept by the fast path: norm^2 > 1e-12


# ---- shard merge (the step after the RCCL all-gather) ---------------------------------------------
@pytest.mark.parametrize("metric", [SCAN_COSINE, SCAN_L2])
@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_sharded_merge_equals_single_device_and_oracle(acc, oracle, n_shards, metric):
    import torch
    n, d, nq, k = 24000, 64, 5, 20
    corpus = oracle.synth_rows(14, 0, n, d)
    q = oracle.synth_rows(14, 1 << 40, nq, d)
    corpus[5] = corpus[23000]                  # a cross-shard exact tie
    q[0] = corpus[5]
    thr = 0.05 if metric == SCAN_L2 else -1.0
    tq = torch.from_numpy(q).cuda()
    bounds = [n * i // n_shards for i in range(n_shards + 1)]
    S = torch.empty((n_shards, nq, k), dtype=torch.float32, device="cuda")
    R = torch.empty((n_shards, nq, k), dtype=torch.int64, device="cuda")
    Cn = torch.empty((n_shards, nq), dtype=torch.int32, device="cuda")
    D = torch.empty((n_shards, nq, k), dtype=torch.float32, device="cuda")
    keep = []
    for s in range(n_shards):
        tc = torch.from_numpy(corpus[bounds[s]:bounds[s + 1]]).cuda(); keep.append(tc)
        view = acc.corpus_view(tc.data_ptr(), bounds[s + 1] - bounds[s], d, row_base=bounds[s])
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, thr, metric, S[s].data_ptr(), R[s].data_ptr(),
                             Cn[s].data_ptr(), D[s].data_ptr(), None,
                             flags=_lib.FLAG_DEFER_THRESHOLD if metric == SCAN_L2 else 0)
    os_ = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    or_ = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    oc = torch.empty(nq, dtype=torch.int32, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    acc.merge_topk_device(n_shards, nq, k, thr, metric, S.data_ptr(), R.data_ptr(), Cn.data_ptr(),
                          D.data_ptr(), None, os_.data_ptr(), or_.data_ptr(), oc.data_ptr(), od.data_ptr())
    acc.synchronize()
    for qi in range(nq):
        if metric == SCAN_COSINE:
            rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, thr)
        else:
            rows, dist, sims = oracle.scan_l2(corpus, q[qi], k, thr)
        c = int(oc[qi])
        assert c == len(rows)
        assert np.array_equal(or_[qi, :c].cpu().numpy(), rows)
        assert np.array_equal(os_[qi, :c].cpu().numpy().view(np.uint32), sims.view(np.uint32))


# ---- sharding behind the C ABI (yams_scan_sharded_*) -------------------------------------------------
@pytest.mark.parametrize("metric", [SCAN_COSINE, SCAN_L2])
@pytest.mark.parametrize("n_shards,layout", [(1, "contiguous"), (2, "contiguous"), (3, "contiguous"), (2, "striped"), (4, "striped")])
def test_sharded_search_behind_one_c_call(oracle, n_shards, layout, metric):
    """One yams_scan_sharded_topk_host call over shards that each have their own context (here all
    on device 0): contiguous row ranges or stripes, cross-shard exact ties, a corpus-wide chunk_id
    ranking through the rank_of_row table; merged result == the oracle over the whole corpus."""
    from yams_amd.accel import ShardedScan
    n, d, nq, k = 30011, 256, 7, 25
    corpus = oracle.synth_rows(41, 0, n, d)
    q = oracle.synth_rows(41, 1 << 40, nq, d)
    corpus[5] = corpus[n - 7] = corpus[n // 2 + 3]        # exact ties across shards
    q[0] = corpus[5]
    rng = np.random.default_rng(5)
    rank = rng.permutation(n).astype(np.uint32)            # the corpus-wide chunk_id ranking
    stripe = 4096
    sh = ShardedScan([0] * n_shards)
    keep, views = [], []
    for i in range(n_shards):
        if layout == "contiguous":
            lo, hi = n * i // n_shards, n * (i + 1) // n_shards
            glob = np.arange(lo, hi)
            extra = dict(row_base=lo)
        else:
            t = np.arange(n) // stripe
            glob = np.flatnonzero(t % n_shards == i)
            extra = dict(stripe_rows=stripe, n_stripes=n_shards, stripe_index=i)
        a = sh.ctx(i)
        part = np.ascontiguousarray(corpus[glob])
        dc = a.to_device(part)
        # local tie ranks: a permutation of 0..n_local-1 that preserves the global order
        order = np.argsort(rank[glob], kind="stable")
        local_rank = np.empty(len(glob), np.uint32); local_rank[order] = np.arange(len(glob), dtype=np.uint32)
        inv = np.empty_like(local_rank); inv[local_rank] = np.arange(len(glob), dtype=np.uint32)
        dr, di = a.to_device(local_rank), a.to_device(inv)
        d8 = dm8 = None
        if metric == SCAN_COSINE:
            d8, dm8 = a.alloc(_lib.i8_shadow_rows(len(glob)) * d), a.alloc((len(glob) + 15) // 16 * 8)
            a.build_shadow_i8_device(dc.ptr, len(glob), d, d8.ptr, dm8.ptr)
        keep += [dc, dr, di, d8, dm8]
        views.append(a.corpus_view(dc.ptr, len(glob), d, dr.ptr, di.ptr, rows_i8_ptr=d8.ptr if d8 else None,
                                   rows_i8_meta_ptr=dm8.ptr if dm8 else None, **extra))
    drank = sh.ctx(0).to_device(rank)
    thr = 0.05 if metric == SCAN_L2 else -1.0
    r = sh.topk(views, q, k, thr, metric, rank_of_row_ptr=drank.ptr)
    for qi in range(nq):
        if metric == SCAN_COSINE:
            rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, thr, rank.astype(np.uint64))
            dist = None
        else:
            rows, dist, sims = oracle.scan_l2(corpus, q[qi], k, thr, rank.astype(np.uint64))
        c = int(r.counts[qi])
        assert c == len(rows) and np.array_equal(r.rows[qi, :c], rows), (qi, r.rows[qi, :8], rows[:8])
        assert np.array_equal(r.scores[qi, :c].view(np.uint32), sims.view(np.uint32))
        if dist is not None:
            assert np.array_equal(r.dist[qi, :c].view(np.uint32), dist.view(np.uint32))
    assert r.diag["rows_visited"] == nq * n
    # invalid query: the batch fails as a whole
    bad = q.copy(); bad[3, 1] = np.nan                  # (a zero query is invalid for cosine only: vec0 accepts it)
    with pytest.raises(_lib.AccelError) as e:
        sh.topk(views, bad, k, thr, metric, rank_of_row_ptr=drank.ptr)
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    sh.close()


@pytest.mark.parametrize("metric", [SCAN_COSINE, SCAN_L2])
def test_small_corpus_runs_as_one_fused_launch(acc, oracle, metric):
    """Corpora of at most 16384 rows with at most 16 queries (BASELINE config 1 and the reference's one-query calls,
    search_vector_pipeline.cpp:221): scan_small_kernel scores every row in fp64 in the reference's order and reduces to
    the final top-k in ONE launch.  Against the oracle bit for bit: 1 / 4 / 16 queries, ragged row counts around the
    256-row workgroups, tie ranks, allow-masks, thresholds, a shard with a row base and with stripes, hostile rows,
    the record path's zero-norm rule; and against the exhaustive multi-launch pipeline (FORCE_EXACT)."""
    rng = np.random.default_rng(91)
    for n, d, nq, k in [(10_000, 384, 1, 10), (10_000, 384, 16, 10), (257, 64, 3, 5), (1, 32, 1, 3), (255, 96, 2, 256),
                        (16_384, 128, 4, 16), (5000, 768, 7, 50), (3000, 1024, 2, 17)]:
        corpus = oracle.synth_rows(61, 0, n, d)
        q = oracle.synth_rows(61, 1 << 40, nq, d)
        if n > 600:
            corpus[5] = corpus[n - 2] = corpus[n // 2]         # exact ties
            q[0] = corpus[5]
            corpus[7] = 0.0                                     # zero row: skipped by cosine, a valid L2 neighbour
            corpus[9] = np.float32(3e38) / 4                    # +-FLT_MAX/4 (vector_smoke_catch2_test.cpp:263-302)
            corpus[11, 3] = np.nan
        thr = 0.02 if metric == SCAN_L2 else -1.0
        rank = rng.permutation(n).astype(np.uint32)
        inv = np.empty_like(rank); inv[rank] = np.arange(n, dtype=np.uint32)
        dc, dr, di = acc.to_device(corpus), acc.to_device(rank), acc.to_device(inv)
        for use_rank in (False, True):
            view = acc.corpus_view(dc.ptr, n, d, dr.ptr if use_rank else None, di.ptr if use_rank else None, row_base=1000)
            r = acc.scan_topk(view, q, k, thr, metric)
            assert r.diag["path"] == 1 and r.diag["filter_tier"] == 0 and r.diag["rows_visited"] == nq * n
            rk = rank.astype(np.uint64) if use_rank else None
            for qi in range(nq):
                if metric == SCAN_COSINE:
                    rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, thr, rk)
                    dist = None
                else:
                    rows, dist, sims = oracle.scan_l2(corpus, q[qi], k, thr, rk)
                c = int(r.counts[qi])
                assert c == len(rows) and np.array_equal(r.rows[qi, :c], rows + 1000), (n, d, nq, k, qi, r.rows[qi, :6], rows[:6])
                assert np.array_equal(r.scores[qi, :c].view(np.uint32), sims.view(np.uint32))
                if dist is not None:
                    assert np.array_equal(r.dist[qi, :c].view(np.uint32), dist.view(np.uint32))
                else:
                    assert np.array_equal(r.dist[qi, :c], np.float32(1.0) - r.scores[qi, :c])
                assert (r.rows[qi, c:] == -1).all()
            ex = acc.scan_topk(view, q, k, thr, metric, flags=FLAG_FORCE_EXACT)     # the multi-launch exhaustive pipeline
            assert np.array_equal(ex.rows, r.rows) and np.array_equal(ex.counts, r.counts)
            assert np.array_equal(ex.scores.view(np.uint32), r.scores.view(np.uint32))
    # allow-masks (document_hash / candidate_hashes), dense and sparse, with tie ranks
    n, d = 9000, 128
    corpus = oracle.synth_rows(62, 0, n, d)
    q = oracle.synth_rows(62, 1 << 40, 5, d)
    rank = rng.permutation(n).astype(np.uint32)
    for allowed in (np.sort(rng.choice(n, 6000, replace=False)), np.sort(rng.choice(n, 40, replace=False)), []):
        r = _masked(acc, oracle, corpus, q, 12, allowed, metric, tie_rank=rank)
        assert r.diag["path"] == 1
    # a striped shard: local rows -> global ids
    n, d = 8192 + 300, 64
    corpus = oracle.synth_rows(63, 0, n, d)
    q = oracle.synth_rows(63, 1 << 40, 2, d)
    dc = acc.to_device(corpus)
    view = acc.corpus_view(dc.ptr, n, d, row_base=7, stripe_rows=4096, n_stripes=3, stripe_index=1)
    r = acc.scan_topk(view, q, 9, -1.0, metric)
    for qi in range(2):
        rows = (oracle.scan_cosine(corpus, q[qi], 9)[0] if metric == SCAN_COSINE else oracle.scan_l2(corpus, q[qi], 9, -1.0)[0])
        glob = 7 + ((rows // 4096) * 3 + 1) * 4096 + rows % 4096
        assert np.array_equal(r.rows[qi], glob)
    # invalid queries: the batch fails as a whole (:4127-4130); a zero query is invalid for cosine only
    bad = q.copy(); bad[1, 5] = np.inf
    with pytest.raises(_lib.AccelError) as e:
        acc.scan_topk(view, bad, 9, -1.0, metric)
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    zq = q.copy(); zq[0] = 0.0
    if metric == SCAN_COSINE:
        with pytest.raises(_lib.AccelError):
            acc.scan_topk(view, zq, 9, -1.0, metric)
    else:
        assert acc.scan_topk(view, zq, 9, -1.0, metric).counts[0] == 9
    # and the handle is fine afterwards (the ticket counters were left at zero)
    assert acc.scan_topk(view, q, 9, -1.0, metric).counts.tolist() == [9, 9]


def test_small_corpus_record_path_zero_norm_rule(acc, oracle):
    """YAMS_SCAN_FLAG_RECORD_PATH on the fused path: rows with norm^2 in [1e-12, 1e-10) are dropped (:204-211)."""
    n, d = 4000, 64
    corpus = oracle.synth_rows(64, 0, n, d)
    q = oracle.synth_rows(64, 1 << 40, 3, d)
    qu = q / np.linalg.norm(q, axis=1, keepdims=True)
    for i in range(3):
        corpus[100 + i] = (qu[i] * 5e-6).astype(np.float32)      # norm^2 = 2.5e-11, cosine 1.0 with query i
    dc = acc.to_device(corpus)
    view = acc.corpus_view(dc.ptr, n, d)
    fast = acc.scan_topk(view, q, 5, -1.0, SCAN_COSINE)
    rec = acc.scan_topk(view, q, 5, -1.0, SCAN_COSINE, flags=_lib.FLAG_RECORD_PATH)
    assert fast.diag["path"] == 1 and rec.diag["path"] == 1
    allow = np.ones(n, np.uint8)
    for qi in range(3):
        assert fast.rows[qi, 0] == 100 + qi and rec.rows[qi, 0] != 100 + qi
        rows, sims = oracle.scan_cosine_records(corpus, q[qi], 5, -1.0, None, allow)[:2]
        assert np.array_equal(rec.rows[qi], rows) and np.array_equal(rec.scores[qi].view(np.uint32), sims.view(np.uint32))


def _one_shard_views(sh, oracle, corpus, d, parts):
    """Uploads `corpus` as len(parts) contiguous shards through the handle's contexts (int8 + bf16 shadows)."""
    keep, views = [], []
    for i, (lo, hi) in enumerate(parts):
        a = sh.ctx(i)
        part = np.ascontiguousarray(corpus[lo:hi])
        dc = a.to_device(part)
        db, dn = a.alloc((hi - lo) * d * 2), a.alloc((hi - lo) * 4)
        a.build_shadow_device(dc.ptr, hi - lo, d, db.ptr, dn.ptr)
        d8, dm8 = a.alloc(_lib.i8_shadow_rows(hi - lo) * d), a.alloc((hi - lo + 15) // 16 * 8)
        a.build_shadow_i8_device(dc.ptr, hi - lo, d, d8.ptr, dm8.ptr)
        a.synchronize()                                 # (the lanes search on their own streams)
        keep += [dc, db, dn, d8, dm8]
        views.append(a.corpus_view(dc.ptr, hi - lo, d, row_base=lo, rows_bf16_ptr=db.ptr, rows_nsq_ptr=dn.ptr,
                                   rows_i8_ptr=d8.ptr, rows_i8_meta_ptr=dm8.ptr))
    return keep, views


def _check_vs_oracle(oracle, corpus, q, r, k, thr, metric):
    for qi in range(q.shape[0]):
        if metric == SCAN_COSINE:
            rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, thr)
            dist = None
        else:
            rows, dist, sims = oracle.scan_l2(corpus, q[qi], k, thr)
        c = int(r.counts[qi])
        assert c == len(rows) and np.array_equal(r.rows[qi, :c], rows), (qi, r.rows[qi, :8], rows[:8])
        assert np.array_equal(r.scores[qi, :c].view(np.uint32), sims.view(np.uint32))
        if dist is not None:
            assert np.array_equal(r.dist[qi, :c].view(np.uint32), dist.view(np.uint32))


def test_sharded_search_through_an_rccl_communicator_of_one_rank(oracle):
    """The C-ABI sharded path with its collective REQUIRED: ncclCommInitAll over the handle's devices (here one:
    what a one-GPU box can run of RCCL), per batch one ncclAllGather of the packed record on the lane's side
    stream, merge_topk_kernel behind it, the download into pinned memory — two batches in flight on two lanes,
    both metrics, oracle-checked bit for bit.  A batch that fails (NaN query) still takes part in its
    collective, so the handle stays usable."""
    from yams_amd.accel import ShardedScan
    n, d, k = 30011, 256, 25
    corpus = oracle.synth_rows(43, 0, n, d)
    qa = oracle.synth_rows(43, 1 << 40, 9, d)
    qb = oracle.synth_rows(43, (1 << 40) + 100, 5, d)
    sh = ShardedScan([0], lanes=2, collective="rccl")
    info = sh.info()
    assert info["collective"] == "rccl" and info["communicator_ranks"] == 1 and info["rccl_version"] > 20000, info
    assert "rccl" in info["rccl_library"]
    keep, views = _one_shard_views(sh, oracle, corpus, d, [(0, n)])
    la = sh.submit(views, qa, k, -1.0, SCAN_COSINE)
    lb = sh.submit(views, qb, k, 0.05, SCAN_L2)
    assert la != lb
    with pytest.raises(_lib.AccelError) as e:       # both lanes have a batch in flight
        sh.submit(views, qa, k, -1.0, SCAN_COSINE, block=False)
    assert e.value.status == _lib.YAMS_ERR_NOT_FOUND
    ra, rb = sh.wait(la), sh.wait(lb)
    _check_vs_oracle(oracle, corpus, qa, ra, k, -1.0, SCAN_COSINE)
    _check_vs_oracle(oracle, corpus, qb, rb, k, 0.05, SCAN_L2)
    assert ra.diag["rows_visited"] == 9 * n and ra.diag["filter_tier"] in (_lib.TIER_I8, _lib.TIER_BF16)
    # a failing batch between two good ones
    bad = qa.copy(); bad[2, 7] = np.nan
    l1 = sh.submit(views, bad, k, -1.0, SCAN_COSINE)
    l2 = sh.submit(views, qb, k, -1.0, SCAN_COSINE)
    with pytest.raises(_lib.AccelError) as e:
        sh.wait(l1)
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    _check_vs_oracle(oracle, corpus, qb, sh.wait(l2), k, -1.0, SCAN_COSINE)
    r3 = sh.topk(views, qa, k, -1.0, SCAN_COSINE)   # submit + wait behind one call
    _check_vs_oracle(oracle, corpus, qa, r3, k, -1.0, SCAN_COSINE)
    info = sh.info()
    assert info["batches"] == 5 and info["collectives"] == 5, info
    # empty results before the query is looked at: k == 0
    r0 = sh.topk(views, bad, 0, -1.0, SCAN_COSINE)
    assert (r0.counts == 0).all()
    sh.close()


def test_sharded_pipeline_on_one_device_equals_the_single_call(oracle):
    """Three shards that share device 0 (no communicator possible: RCCL refuses two ranks on one device — asking
    for it is an error), three lanes: three different batches in flight, every result equal to the oracle over
    the whole corpus and to the one-call form; lanes are reused."""
    from yams_amd.accel import ShardedScan
    with pytest.raises(_lib.AccelError) as e:
        ShardedScan([0, 0], collective="rccl")
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    n, d, k = 40000, 256, 20
    corpus = oracle.synth_rows(44, 0, n, d)
    corpus[17] = corpus[n - 3] = corpus[n // 3 + 1]      # exact ties across shards
    batches = [oracle.synth_rows(44, (1 << 40) + 50 * j, 3 + 2 * j, d) for j in range(5)]
    batches[0][0] = corpus[17]
    sh = ShardedScan([0, 0, 0], lanes=3)
    assert sh.info()["collective"] == "peer_copy" and sh.lanes == 3
    parts = [(n * i // 3, n * (i + 1) // 3) for i in range(3)]
    keep, views = _one_shard_views(sh, oracle, corpus, d, parts)
    lanes = [sh.submit(views, b, k, -1.0, SCAN_COSINE) for b in batches[:3]]
    assert sorted(lanes) == [0, 1, 2]
    res = [sh.wait(l) for l in lanes]
    lanes = [sh.submit(views, b, k, -1.0, SCAN_COSINE) for b in batches[3:]]
    res += [sh.wait(l) for l in lanes]
    for b, r in zip(batches, res):
        _check_vs_oracle(oracle, corpus, b, r, k, -1.0, SCAN_COSINE)
        one = sh.topk(views, b, k, -1.0, SCAN_COSINE)
        assert np.array_equal(one.rows, r.rows) and np.array_equal(one.scores.view(np.uint32), r.scores.view(np.uint32))
    sh.close()


@pytest.mark.parametrize("lanes,flag", [(1, _lib.FLAG_L2_ACC_F32), (8, _lib.FLAG_L2_ACC_F32X8), (16, _lib.FLAG_L2_ACC_F32X16),
                                        (-1, _lib.FLAG_L2_ACC_F32 | _lib.FLAG_L2_ACC_FUSED), (-8, _lib.FLAG_L2_ACC_F32X8 | _lib.FLAG_L2_ACC_FUSED),
                                        (-16, _lib.FLAG_L2_ACC_F32X16 | _lib.FLAG_L2_ACC_FUSED)])
def test_l2_under_fp32_accumulation_matches_that_oracle(acc, oracle, lanes, flag):
    """vec0's distance arithmetic lives in the absent sqlite-vec-cpp, so the host picks it
    (YAMS_SCAN_FLAG_L2_ACC_*).  Under fp32 accumulation — sequential, 8 or 16 round-robin lanes — rows, order, distances
    (bits) and the cosine re-score equal oracle_exact_scan_l2_f32acc(lanes) on every tier: int8 (resident and half-tile
    forms), bf16, the exhaustive fp64-free pipeline, small corpora (the fused kernel is fp64-only and steps aside),
    dimensions with tails (100: float4 walk + tail, 37: unaligned scalar walk), hostile rows (fp32 overflow -> inf ->
    skipped, zero rows, NaN), exact ties broken by chunk_id, the threshold applied after the top-k.  Reference:
    sqlite_vec_backend.cpp:4464-4512.  Negative lanes (round 5): the same lanes with every square accumulated by ONE fused
    multiply-add (YAMS_SCAN_FLAG_L2_ACC_FUSED) — what the reference's x86 build of its dependency ('-mavx', '-mfma',
    src/vector/meson.build:80-88) makes of those loops."""
    rng = np.random.default_rng(200 + abs(lanes) + (50 if lanes < 0 else 0))
    cases = [dict(n=60_000, d=256, nq=140, k=20),                                   # int8 tier
             dict(n=60_000, d=256, nq=140, k=20, flags=_lib.FLAG_RESIDENT_QUERIES),  # ... resident-query form
             dict(n=30_011, d=256, nq=7, k=25, thr=0.05),                            # narrow bf16 form, threshold after top-k
             dict(n=20_000, d=100, nq=5, k=10, hostile=True),                        # float4 walk with a tail
             dict(n=9_000, d=37, nq=3, k=10, hostile=True),                          # unaligned rows: scalar walk, exhaustive keys
             dict(n=5_000, d=384, nq=3, k=10, use_rank=True),                        # small corpus: the fused kernel steps aside
             dict(n=40_000, d=768, nq=9, k=100, flags=FLAG_FORCE_EXACT, hostile=True, use_rank=True)]
    for c in cases:
        n, d, nq, k = c["n"], c["d"], c["nq"], c["k"]
        thr, flags = c.get("thr", -1.0), c.get("flags", 0)
        corpus = oracle.synth_rows(77, 0, n, d) * np.float32(rng.uniform(0.5, 2.0))
        q = oracle.synth_rows(77, 1 << 40, nq, d)
        corpus[5] = corpus[n - 2] = corpus[n // 2]                 # exact ties
        q[0] = corpus[5] + np.float32(1e-3)
        if c.get("hostile"):
            corpus[7] = 0.0
            corpus[9] = np.float32(3e38) / 4                       # (x - q)^2 overflows fp32: distance inf, row skipped
            corpus[11, 3] = np.nan
            corpus[13] *= np.float32(1e-22)                        # squares underflow to zero / subnormals
        rank = rng.permutation(n).astype(np.uint32) if c.get("use_rank") else None
        r = run(acc, corpus, q, k, thr, SCAN_L2, flags | flag, tie_rank=rank, shadow="both")
        for qi in (range(nq) if nq <= 9 else (0, 1, nq // 2, nq - 1)):
            rows, dist, sims = oracle.scan_l2_f32acc(corpus, q[qi], k, thr, rank.astype(np.uint64) if rank is not None else None, lanes=lanes)
            cnt = int(r.counts[qi])
            assert cnt == len(rows) and np.array_equal(r.rows[qi, :cnt], rows), (c, lanes, qi, r.rows[qi, :6], rows[:6])
            assert np.array_equal(r.dist[qi, :cnt].view(np.uint32), dist.view(np.uint32)), (c, lanes, qi)
            assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), (c, lanes, qi)
    # the default (no flag) is still the fp64 definition
    corpus = oracle.synth_rows(78, 0, 20_000, 128); q = oracle.synth_rows(78, 1 << 40, 4, 128)
    r = run(acc, corpus, q, 10, -1.0, SCAN_L2)
    for qi in range(4):
        rows, dist, _ = oracle.scan_l2(corpus, q[qi], 10, -1.0)
        assert np.array_equal(r.rows[qi], rows) and np.array_equal(r.dist[qi].view(np.uint32), dist.view(np.uint32))


def test_plugin_serves_the_l2_arithmetic_its_config_names(accel_lib, oracle):
    """{"l2_accumulate": "f32x8"}: every vec0 search of the plugin uses that arithmetic (a call may still name its own)."""
    L = accel_lib
    vt = _vt(L, b'{"device": 0, "l2_accumulate": "f32x8_fma"}')
    n, d, k = 20_000, 256, 10
    corpus = oracle.synth_rows(80, 0, n, d) * np.float32(1.3)
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    assert vt.corpus_append(None, cid, corpus.ctypes.data_as(_lib.f32p), n) == 0
    q = oracle.synth_rows(80, 1 << 40, 3, d)
    rows, _, _ = _vt_search(vt, cid, np.ascontiguousarray(q), k, metric=1)
    for qi in range(3):
        assert rows[qi] == list(oracle.scan_l2_f32acc(corpus, q[qi], k, -1.0, lanes=-8)[0]), qi
    hp = C.c_void_p()
    assert L.yams_plugin_get_health_json(C.byref(hp)) == 0 and json.loads(C.string_at(hp))["l2_accumulate"] == "f32x8_fma"
    C.CDLL(None).free(hp)
    vt = _vt(L, b'{"device": 0, "l2_accumulate": "f32x8"}')
    n, d, k = 50_000, 256, 10
    corpus = oracle.synth_rows(79, 0, n, d) * np.float32(1.7)
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    assert vt.corpus_append(None, cid, corpus.ctypes.data_as(_lib.f32p), n) == 0
    q = oracle.synth_rows(79, 1 << 40, 130, d)
    for nq in (2, 130):
        rows, _, _ = _vt_search(vt, cid, np.ascontiguousarray(q[:nq]), k, metric=1)
        for qi in (0, nq - 1):
            assert rows[qi] == list(oracle.scan_l2_f32acc(corpus, q[qi], k, -1.0, lanes=8)[0]), (nq, qi)
    rows, _, _ = _vt_search(vt, cid, np.ascontiguousarray(q[:2]), k, metric=1, flags=_lib.FLAG_L2_ACC_F32)
    assert rows[0] == list(oracle.scan_l2_f32acc(corpus, q[0], k, -1.0, lanes=1)[0])
    hp = C.c_void_p()
    assert L.yams_plugin_get_health_json(C.byref(hp)) == 0 and json.loads(C.string_at(hp))["l2_accumulate"] == "f32x8"
    C.CDLL(None).free(hp)
    assert vt.corpus_destroy(None, cid) == 0
    L.yams_plugin_shutdown()


def test_plugin_chooses_the_int8_layout_per_corpus(accel_lib, oracle):
    """{"i8_layout": "auto"} (the default): a corpus of rows with outlier dimensions gets the rotated int8 shadow at its first
    append, a corpus of uniform rows the plain one, in the same plugin; "plain" / "rotated" fix it.  Searches of batches that
    take the int8 tier return the oracle's rows either way."""
    L = accel_lib
    n, d, k, nq = 60_000, 256, 20, 140
    rng = np.random.default_rng(81)
    out = rng.standard_normal((n, d)).astype(np.float32); out[:, [3, 100, 200]] *= np.float32(15.0)
    uni = oracle.synth_rows(81, 0, n, d)
    qo = rng.standard_normal((nq, d)).astype(np.float32); qo[:, [3, 100, 200]] *= np.float32(15.0)
    qu = oracle.synth_rows(81, 1 << 40, nq, d)

    def health():
        hp = C.c_void_p()
        assert L.yams_plugin_get_health_json(C.byref(hp)) == 0
        h = json.loads(C.string_at(hp)); C.CDLL(None).free(hp)
        return h

    for cfg, want in ((b'{"device": 0}', (1, 1)), (b'{"device": 0, "i8_layout": "plain"}', (0, 0)), (b'{"device": 0, "i8_layout": "rotated"}', (1, 2)),
                      (b'{"devices": [0, 0], "stripe_rows": 4096, "i8_layout": "rotated"}', (1, 2)), (b'{"devices": [0, 0, 0], "stripe_rows": 8192}', (1, 1))):
        vt = _vt(L, cfg)
        ids = []
        for rows_, qs_, after in ((out, qo, want[0]), (uni, qu, want[1])):
            cid = C.c_uint64()
            assert vt.corpus_create(None, d, C.byref(cid)) == 0
            assert vt.corpus_append(None, cid, rows_[:n // 2].ctypes.data_as(_lib.f32p), n // 2) == 0
            assert vt.corpus_append(None, cid, np.ascontiguousarray(rows_[n // 2:]).ctypes.data_as(_lib.f32p), n - n // 2) == 0   # (appends keep the layout)
            assert health()["corpora_with_rotated_i8_shadow"] == after, (cfg, health())
            got, _, _ = _vt_search(vt, cid, np.ascontiguousarray(qs_), k)
            for qi in (0, 77, nq - 1):
                assert got[qi] == list(oracle.scan_cosine(rows_, qs_[qi], k, -1.0)[0]), (cfg, qi)
            ids.append(cid)
        assert health()["i8_layout"] == ("auto" if b"i8_layout" not in cfg else cfg.split(b'"i8_layout": "')[1].split(b'"')[0].decode())
        for cid in ids:
            assert vt.corpus_destroy(None, cid) == 0
        L.yams_plugin_shutdown()


def test_plugin_measures_the_layout_again_once_enough_rows_are_there(accel_lib, oracle):
    """A host that inserts a handful of vectors and searches before the bulk arrives: the layout measured on 64 rows (uniform
    here: plain) is measured again when the corpus passes 4096 rows (outlier-dimension rows now: rotated) and the whole shadow
    is rebuilt in it; searches before and after return the oracle's rows."""
    L = accel_lib
    vt = _vt(L, b'{"device": 0}')
    d, k = 256, 10
    rng = np.random.default_rng(83)
    head = oracle.synth_rows(83, 0, 64, d)
    bulk = rng.standard_normal((50_000, d)).astype(np.float32); bulk[:, [5, 77, 200]] *= np.float32(15.0)
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    assert vt.corpus_append(None, cid, head.ctypes.data_as(_lib.f32p), 64) == 0

    def rotated():
        hp = C.c_void_p()
        assert L.yams_plugin_get_health_json(C.byref(hp)) == 0
        h = json.loads(C.string_at(hp)); C.CDLL(None).free(hp)
        return h["corpora_with_rotated_i8_shadow"]
    assert rotated() == 0
    got, _, _ = _vt_search(vt, cid, np.ascontiguousarray(head[:3]), k)
    assert got[0] == list(oracle.scan_cosine(head, head[0], k, -1.0)[0])
    assert vt.corpus_append(None, cid, bulk.ctypes.data_as(_lib.f32p), bulk.shape[0]) == 0
    assert rotated() == 1
    allrows = np.concatenate([head, bulk])
    qs = np.ascontiguousarray(np.concatenate([bulk[:140] * np.float32(0.5), head[:4]]))
    got, _, _ = _vt_search(vt, cid, qs, k)
    for qi in (0, 77, 139, 141):
        assert got[qi] == list(oracle.scan_cosine(allrows, qs[qi], k, -1.0)[0]), qi
    assert vt.corpus_destroy(None, cid) == 0
    L.yams_plugin_shutdown()


def test_l2_definition_gap_measured_from_the_device_result(acc, oracle, capsys):
    """L2 parity is unpinned (the vec0 arithmetic lives in the absent sqlite-vec-cpp): how much does the fp64-vs-fp32
    accumulation choice matter?  The device returns the top 200 under this repository's definition; the fp32-accumulated
    distances of those candidates (sequential, 8 and 16 SIMD lanes) give the top-100 under the other definitions.
    Asserted: distances agree within north_star's 1e-5 (relative) and the method is conclusive; REPORTED: the sets
    that differ."""
    import torch
    n, d, nq, k, K = 400_000, 768, 64, 100, 200
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
    s = torch.empty((nq, K), dtype=torch.float32, device="cuda"); r = torch.empty((nq, K), dtype=torch.int64, device="cuda")
    c = torch.empty(nq, dtype=torch.int32, device="cuda"); dist = torch.empty((nq, K), dtype=torch.float32, device="cuda")
    view = acc.corpus_view(tc.data_ptr(), n, d)
    acc.scan_topk_device(view, tq.data_ptr(), nq, K, -1.0, SCAN_L2, s.data_ptr(), r.data_ptr(), c.data_ptr(), dist.data_ptr())
    assert (c == K).all()
    rep = _oracle.l2_definition_report(oracle, lambda rows: tc[torch.from_numpy(rows).cuda()].cpu().numpy(), tq.cpu().numpy(),
                                       r.cpu().numpy(), dist.cpu().numpy(), k)
    for name, v in rep["variants"].items():
        assert v["max_rel_distance_difference"] < 1e-5, (name, v)
        assert v["queries_not_conclusive"] == 0, (name, v)
    with capsys.disabled():
        print("\n[L2 definition report, device candidates] " + json.dumps(rep))


# ---- BASELINE.json full sizes: size-independent properties ----------------------------------------
def _full_size(acc, oracle, n, d, nq, k, metric, n_oracle_queries):
    import torch
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda")
    acc.synth_rows(42, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda")
    acc.synth_rows(42, n, nq, d, tq.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c = torch.empty(nq, dtype=torch.int32, device="cuda")
    dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")
    tn = torch.empty(n, dtype=torch.float32, device="cuda")
    acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
    acc.synchronize()
    # the shadow is the RNE bf16 of the unit-normalised rows, plus their squared norms
    n64 = (tc[:4096].double() ** 2).sum(-1)
    assert torch.allclose(tn[:4096], n64.float(), rtol=1e-5)
    unit = (tc[:4096].double() / n64.sqrt()[:, None]).float()
    assert (tb[:4096].float() - unit).abs().max().item() <= 2.0 ** -8 * unit.abs().max().item() * 1.01
    t8 = tm8 = None
    if (metric == SCAN_COSINE and d % 64 == 0 and d >= 256) or (d % 128 == 0 and 256 <= d <= 768):   # the int8 shadow: first filter tier of cosine batches > 128 queries, and of L2 batches on shards that take the resident-query form
        t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda")
        tm8 = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
        mean_err = acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr(), want_mean_err=True)
        assert 0.0 < mean_err < 0.006                      # uniform components: ~ sqrt(3) / (127 sqrt(12)) = 0.0039
        unit8 = (tc[:4096].double() / n64.sqrt()[:, None])
        sc8 = tm8[:64, 0].double().repeat_interleave(64)[:, None]        # one scale per block of 64 rows
        recon = unblock_i8_shadow(t8[:4096], d).double() * sc8
        assert ((unit8 - recon).norm(dim=-1) <= tm8[:64, 1].double().repeat_interleave(64)).all()    # e_b bounds the measured residues
    view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                           rows_i8_ptr=t8.data_ptr() if t8 is not None else None,
                           rows_i8_meta_ptr=tm8.data_ptr() if tm8 is not None else None)
    diag = acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, metric, s.data_ptr(), r.data_ptr(),
                                c.data_ptr(), dist.data_ptr())
    assert diag["path"] == 0 and diag["exact_fallback_queries"] == 0
    assert diag["filter_tier"] == (_lib.TIER_I8 if t8 is not None else _lib.TIER_BF16)
    if t8 is not None:                                     # the bf16 tier gives the same bits
        sb = torch.empty_like(s); rb_ = torch.empty_like(r); cb = torch.empty_like(c)
        dg = acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, metric, sb.data_ptr(), rb_.data_ptr(),
                                  cb.data_ptr(), None, None, flags=_lib.FLAG_NO_I8_FILTER)
        assert dg["filter_tier"] == _lib.TIER_BF16 and torch.equal(rb_, r) and torch.equal(sb, s) and torch.equal(cb, c)
    assert (c == k).all()
    key = dist if metric == SCAN_L2 else -s
    assert (key[:, 1:] >= key[:, :-1]).all()                                 # sortedness
    assert ((r >= 0) & (r < n)).all()
    assert all(len(set(row.tolist())) == k for row in r.cpu().numpy()[:32])  # no duplicates
    # the fp64 scores of the returned rows, recomputed independently with torch on the device
    sub = slice(0, 8)
    rows = tc[r[sub].reshape(-1)].double().reshape(8, k, d)
    qd = tq[sub].double()
    if metric == SCAN_COSINE:
        ref = (rows * qd[:, None, :]).sum(-1) / (rows.norm(dim=-1) * qd.norm(dim=-1)[:, None])
        assert (ref.float() - s[sub]).abs().max().item() <= 1e-6
    else:
        ref = (rows - qd[:, None, :]).norm(dim=-1)
        assert (ref.float() - dist[sub]).abs().max().item() <= 1e-5
    # recall@k = 1.0 against the exhaustive fp64 device path (itself pinned to the oracle above)
    qsel = torch.arange(0, nq, max(1, nq // 4))[:4].cuda()
    s2 = torch.empty((len(qsel), k), dtype=torch.float32, device="cuda")
    r2 = torch.empty((len(qsel), k), dtype=torch.int64, device="cuda")
    c2 = torch.empty(len(qsel), dtype=torch.int32, device="cuda")
    tq2 = tq[qsel].contiguous()
    acc.scan_topk_device(view, tq2.data_ptr(), len(qsel), k, -1.0, metric, s2.data_ptr(), r2.data_ptr(),
                         c2.data_ptr(), None, None, flags=FLAG_FORCE_EXACT)
    assert torch.equal(r2, r[qsel]) and torch.equal(s2, s[qsel])
    # and against the CPU oracle (all host cores, row slices of the very tensor the GPU scanned,
    # comparator merge: tests/_oracle.py scan_threaded == one oracle call over the whole corpus)
    if n_oracle_queries:
        if n_oracle_queries == "all":   # EVERY query of the batch (the batched oracle drivers: tests/_oracle.py, MANY_FROM)
            n_oracle_queries = nq
        qsel_o = [int(x) for x in np.linspace(0, nq - 1, n_oracle_queries).round()]
        queries = tq[qsel_o].cpu().numpy()
        # the device generator is the oracle's Philox recipe: spot-check a slice at the far end
        lo = max(0, n - 1000)
        assert np.array_equal(tc[lo:n].cpu().numpy().view(np.uint32), oracle.synth_rows(42, lo, n - lo, d).view(np.uint32))
        ref = _oracle.scan_threaded(lambda a, b: tc[a:b].cpu().numpy(), n, queries, k,
                                    metric="l2" if metric == SCAN_L2 else "cosine", thr=-1.0)
        rr, ss, dd = r.cpu().numpy(), s.cpu().numpy(), dist.cpu().numpy()
        for j, qi in enumerate(qsel_o):
            assert np.array_equal(rr[qi], ref[j][0]), (qi, rr[qi][:8], ref[j][0][:8])
            assert np.array_equal(ss[qi].view(np.uint32), ref[j][1].view(np.uint32)), qi
            if metric == SCAN_L2:
                assert np.array_equal(dd[qi].view(np.uint32), ref[j][2].view(np.uint32)), qi


def test_full_size_config2_1Mx384_cosine_top100_q256(acc, oracle):
    """BASELINE config 2, all 256 queries against the oracle over all 1M rows."""
    _full_size(acc, oracle, 1_000_000, 384, 256, 100, SCAN_COSINE, n_oracle_queries="all")


def test_full_size_config3_10Mx768_l2_top100_q1024(acc, oracle):
    """BASELINE config 3, all 1024 queries against the oracle over all 10M rows (the oracle's L2 is THIS repository's
    fp64 definition: parity unpinned, oracle/yams_oracle.c)."""
    _full_size(acc, oracle, 10_000_000, 768, 1024, 100, SCAN_L2, n_oracle_queries="all")


def test_full_size_config4_shard_12p5Mx768_cosine_top100_q1024(acc, oracle):
    """One row shard of BASELINE config 4 (100M x 768 over 8 GPUs) = the bench.py workload: the int8-shadow filter at
    Q = 1024 over 12.5M rows, EVERY one of the 1024 queries against the oracle over the full shard (rows, order, score
    bits) — about a minute of the host's cores."""
    _full_size(acc, oracle, 12_500_000, 768, 1024, 100, SCAN_COSINE, n_oracle_queries="all")


def test_full_size_config4_shard_small_batch_takes_the_int8_stream(acc, oracle):
    """The same shard with 8 queries: on a shard this large small batches take the int8 resident-query form with
    one query tile (every CU streams its own rows of the int8 shadow — half the bytes of the narrow bf16 form);
    two queries against the oracle over the full shard, all eight against the bf16 tier and the exhaustive path."""
    _full_size(acc, oracle, 12_500_000, 768, 8, 100, SCAN_COSINE, n_oracle_queries=2)


def test_config1_10kx384_cosine_top10_single_query_reference_recipe(acc, oracle):
    """BASELINE config 1: the reference's own CPU-runnable case — 10 000 x 384, k = 10, one query
    per call, data from ITS recipe (std::mt19937(42), U(-1,1), fp32 normalise; corpus first, then
    the queries from the same stream: tests/benchmarks/vector_backend_engine_compare.cpp:83-107,
    251-253), here through the device path: every query as its own batch, then all in one batch."""
    n, d, k, nq = 10_000, 384, 10, 16
    corpus = oracle.mt19937_rows(42, 0, n, d)
    queries = oracle.mt19937_rows(42, n, nq, d)
    for qi in range(nq):
        r = check(acc, oracle, corpus, queries[qi], k, thr=0.0, expect_path=1)   # the reference bench passes 0.0; one fused launch
        assert r.counts[0] == k and r.diag["exact_fallback_queries"] == 0
    check(acc, oracle, corpus, queries, k, thr=-1.0, expect_path=1)
    check(acc, oracle, corpus, queries, k, thr=-1.0, flags=_lib.FLAG_NO_I8_FILTER, expect_path=0)   # the MFMA filter pipeline on the same shape
    check(acc, oracle, corpus, queries[:1], k, thr=-1.0, flags=FLAG_FORCE_EXACT, expect_path=1)



# ---- the INT8 filter tier ----------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,nq,k,thr", [
    (20000, 256, 1, 10, -1.0), (20007, 320, 5, 50, 0.1), (50000, 256, 130, 100, -1.0),
    (70001, 256, 300, 50, -1.0), (33333, 768, 40, 100, 0.05), (150000, 384, 260, 100, -1.0),
    (9000, 1024, 17, 20, -1.0), (4096, 448, 3, 300, -1.0), (30011, 1536, 33, 10, -1.0),
])
def test_int8_tier_matches_the_oracle(acc, oracle, n, d, nq, k, thr):
    """The int8 tier (v_mfma_i32_32x32x32_i8 over the int8 shadow; filter score = a rigorous upper
    bound from the measured quantisation residues) in front of the same fp64 re-score + proof:
    bit-identical to the oracle; ragged row tails, several query tiles, thresholds."""
    corpus = oracle.synth_rows(31, 0, n, d)
    q = oracle.synth_rows(31, 1 << 40, nq, d)
    r = check(acc, oracle, corpus, q, k, thr=thr, max_queries=10, expect_path=0, shadow="i8", expect_tier=_lib.TIER_I8)
    assert r.diag["exact_fallback_queries"] == 0
    # both shadows: the int8 tier from 129 queries on, the narrow bf16 form below; NO_I8 switches it off
    r2 = check(acc, oracle, corpus, q, k, thr=thr, max_queries=4, expect_path=0, shadow="both",
               expect_tier=_lib.TIER_I8 if nq > 128 else _lib.TIER_BF16)
    r3 = run(acc, corpus, q, k, thr, flags=_lib.FLAG_NO_I8_FILTER, shadow="both")
    assert r3.diag["filter_tier"] == _lib.TIER_BF16
    for other in (r2, r3):
        assert np.array_equal(r.rows, other.rows) and np.array_equal(r.counts, other.counts)
        assert np.array_equal(r.scores.view(np.uint32), other.scores.view(np.uint32))


@pytest.mark.parametrize("n,d,nq,k,thr", [
    (20000, 256, 1, 10, -1.0), (50000, 256, 130, 100, -1.0), (70001, 256, 300, 50, 0.02),
    (33333, 768, 40, 100, 0.05), (150001, 384, 260, 100, -1.0), (40449, 512, 700, 30, -1.0),
    (4500, 640, 129, 20, -1.0), (90000, 768, 1024, 100, -1.0), (25000, 768, 2100, 10, -1.0),
])
def test_int8_resident_query_form_matches_the_oracle_and_the_half_tile_form(acc, oracle, n, d, nq, k, thr):
    """The resident-query form of the int8 filter (scan_tiles_i8r_kernel: a 128-query tile stays in LDS,
    persistent workgroups stream 512-row units past it, wave-private row rings) forced on shards far
    smaller than the ones the library picks it for: ragged last units, odd tile counts, streams without
    work, one to seventeen query tiles, thresholds, an allow-mask — bit-identical to the oracle and to the
    half-tile form (YAMS_SCAN_FLAG_WIDE_TILE keeps the per-tile kernels)."""
    corpus = oracle.synth_rows(35, 0, n, d)
    q = oracle.synth_rows(35, 1 << 40, nq, d)
    r = check(acc, oracle, corpus, q, k, thr=thr, max_queries=8, expect_path=0, shadow="i8", expect_tier=_lib.TIER_I8,
              flags=_lib.FLAG_RESIDENT_QUERIES)
    assert r.diag["exact_fallback_queries"] == 0
    h = run(acc, corpus, q, k, thr, flags=FLAG_WIDE_TILE, shadow="i8")
    assert h.diag["filter_tier"] == _lib.TIER_I8
    assert np.array_equal(r.rows, h.rows) and np.array_equal(r.counts, h.counts)
    assert np.array_equal(r.scores.view(np.uint32), h.scores.view(np.uint32))
    assert r.diag["filter_candidates"] == h.diag["filter_candidates"]     # the same survivors, not just the same top k
    rng = np.random.default_rng(n)
    mask = rng.random(n) < 0.5
    a = run(acc, corpus, q, k, thr, flags=_lib.FLAG_RESIDENT_QUERIES, shadow="i8", mask=mask)
    b = run(acc, corpus, q, k, thr, flags=FLAG_WIDE_TILE, shadow="i8", mask=mask)
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.counts, b.counts)
    assert np.array_equal(a.scores.view(np.uint32), b.scores.view(np.uint32))
    assert a.diag["filter_candidates"] == b.diag["filter_candidates"]


def _l2_corpus(oracle, seed, n, d, nq, kind):
    """Row / query sets for the L2 int8 tier: the synthetic uniform rows as they are, unit rows, norms spread
    within a factor of two (the tier's limit), or tight clusters with the queries next to rows (tau' > 0)."""
    rng = np.random.default_rng(seed)
    corpus = oracle.synth_rows(seed, 0, n, d)
    q = oracle.synth_rows(seed, 1 << 40, nq, d)
    if kind == "unit":
        corpus = (corpus / np.linalg.norm(corpus.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    elif kind == "spread":
        corpus = (corpus * rng.uniform(0.85, 1.35, (n, 1))).astype(np.float32)
    elif kind == "clustered":
        centres = rng.normal(0, 1, (8, d)).astype(np.float32)
        corpus = (centres[rng.integers(0, 8, n)] + 0.05 * rng.normal(0, 1, (n, d))).astype(np.float32)
        q = (corpus[rng.integers(0, n, nq)] + 0.02 * rng.normal(0, 1, (nq, d))).astype(np.float32)
    if nq > 2 and kind != "clustered":
        q[0] *= np.float32(3.0); q[1] *= np.float32(0.2)     # queries of other lengths than the rows
    return np.ascontiguousarray(corpus), np.ascontiguousarray(q)


@pytest.mark.parametrize("n,d,nq,k,thr,kind", [
    (50000, 256, 130, 100, -1.0, "uniform"), (70001, 384, 300, 50, 0.05, "uniform"), (40449, 512, 700, 30, -1.0, "unit"),
    (90000, 768, 1024, 100, -1.0, "spread"), (33333, 768, 40, 100, -1.0, "clustered"), (20500, 640, 129, 20, -1.0, "uniform"),
    (20000, 256, 1, 10, -1.0, "unit"), (60000, 320, 200, 50, -1.0, "uniform"), (30011, 1536, 133, 10, -1.0, "spread"),
    (150001, 448, 260, 100, 0.02, "unit"),
])
def test_int8_tier_under_l2_matches_the_oracle_and_the_bf16_tier(acc, oracle, n, d, nq, k, thr, kind):
    """L2 batches on the int8 tier (scan_i8_kernel.hip, "L2 on the int8 tier"): the cosine tier's shadow, the score
    bound G(u, |x|^2), an integer threshold with a per-row part — in the resident-query form (forced onto these small
    shards through the flag; dims it cannot hold fall to half tiles) and in the half-tile form.  Bit-identical to
    the oracle (rows, cosines, distances) and to the bf16 tier, with and without an allow-mask; nothing escalates on
    well-behaved norms."""
    corpus, q = _l2_corpus(oracle, 71, n, d, nq, kind)
    r = check(acc, oracle, corpus, q, k, thr=thr, metric=SCAN_L2, max_queries=6, expect_path=0, shadow="both",
              expect_tier=_lib.TIER_I8, flags=_lib.FLAG_RESIDENT_QUERIES)
    if kind != "clustered":
        assert r.diag["exact_fallback_queries"] == 0, r.diag
    b = run(acc, corpus, q, k, thr, SCAN_L2, flags=_lib.FLAG_NO_I8_FILTER, shadow="both")
    assert b.diag["filter_tier"] == _lib.TIER_BF16
    h = run(acc, corpus, q, k, thr, SCAN_L2, flags=FLAG_WIDE_TILE, shadow="both")       # the per-tile kernel forms
    if nq > 128:
        assert h.diag["filter_tier"] == _lib.TIER_I8
        assert h.diag["filter_candidates"] == r.diag["filter_candidates"]            # the same survivors, not just the same top k
    for other in (b, h, run(acc, corpus, q, k, thr, SCAN_L2, shadow="both")):
        assert np.array_equal(r.rows, other.rows) and np.array_equal(r.counts, other.counts)
        assert np.array_equal(r.scores.view(np.uint32), other.scores.view(np.uint32))
        assert np.array_equal(r.dist.view(np.uint32), other.dist.view(np.uint32))
    mask = np.random.default_rng(n).random(n) < 0.5
    a = run(acc, corpus, q, k, thr, SCAN_L2, flags=_lib.FLAG_RESIDENT_QUERIES, shadow="both", mask=mask)
    c = run(acc, corpus, q, k, thr, SCAN_L2, flags=_lib.FLAG_NO_I8_FILTER, shadow="both", mask=mask)
    if a.diag["path"] == 0:      # (half of a small shard is a sparse mask: gathered and scored in fp64, no filter tier)
        assert a.diag["filter_tier"] == _lib.TIER_I8 and c.diag["filter_tier"] == _lib.TIER_BF16
    assert np.array_equal(a.rows, c.rows) and np.array_equal(a.counts, c.counts)
    assert np.array_equal(a.dist.view(np.uint32), c.dist.view(np.uint32))


def test_int8_tier_under_l2_rows_without_a_usable_norm(acc, oracle):
    """Zero rows (valid L2 neighbours, but there is no unit vector to quantise), rows whose squared norm overflows
    the filter's range and rows with a non-finite component have no score bound: up to 64 of them ride along as
    unconditional candidates of every query and the batch stays on the int8 tier; more than that, or norms spread
    over more than a factor of two, and the batch steps aside to the bf16 tier.  Same results either way."""
    n, d, nq, k = 30000, 256, 140, 20
    base, q = _l2_corpus(oracle, 73, n, d, nq, "uniform")
    q[5] = np.float32(0.01) * base[20000]                         # a short query: the zero rows are among its nearest
    for how, tier in (("few", _lib.TIER_I8), ("many", _lib.TIER_BF16), ("wide", _lib.TIER_BF16)):
        corpus = base.copy()
        if how == "few":
            corpus[[12345, 64, 29999]] = 0                         # zero rows, one of them the shard's last
            corpus[777] *= np.float32(1e16)                        # |x|^2 far outside the range, still finite
            corpus[4096, 3] = np.nan; corpus[9000, 0] = np.inf     # never returned (vector_database.cpp:1771-1784)
            corpus[128:192] *= np.float32(1.0)                     # (a whole block of ordinary rows next to them)
        elif how == "many":
            corpus[1000:1100] = 0
        else:
            corpus[::7] *= np.float32(2.5)
        for flags in (_lib.FLAG_RESIDENT_QUERIES, FLAG_WIDE_TILE):
            r = check(acc, oracle, corpus, q, k, metric=SCAN_L2, max_queries=8, expect_path=0, shadow="both",
                      expect_tier=tier, flags=flags)
            assert r.diag["exact_fallback_queries"] == 0
        if how == "few":
            assert 12345 in r.rows[5] and 64 in r.rows[5] and 29999 in r.rows[5]      # the zero rows are returned
            mask = np.ones(n, bool); mask[[12345, 777]] = False                       # ... unless the allow-mask hides them
            m = run(acc, corpus, q, k, -1.0, SCAN_L2, flags=_lib.FLAG_RESIDENT_QUERIES, shadow="both", mask=mask)
            e = run(acc, corpus, q, k, -1.0, SCAN_L2, flags=FLAG_FORCE_EXACT, shadow="both", mask=mask)
            assert m.diag["filter_tier"] == _lib.TIER_I8 and 12345 not in m.rows[5] and 64 in m.rows[5]
            assert np.array_equal(m.rows, e.rows) and np.array_equal(m.dist.view(np.uint32), e.dist.view(np.uint32))


def test_int8_tier_proves_small_dense_shards_without_escalating(acc, oracle):
    """A 300k-row shard admits ~16 survivors per wave tile (a 12.5M-row one: 0.6): the survivor log is
    sized from the plan, so the int8 tier proves these queries itself instead of escalating all of them."""
    n, d, nq, k = 300_000, 256, 300, 50
    corpus = oracle.synth_rows(61, 0, n, d)
    queries = oracle.synth_rows(61, 1 << 40, nq, d)
    diag = check(acc, oracle, corpus, queries, k, shadow="i8", max_queries=6, expect_tier=_lib.TIER_I8).diag
    assert diag["escalated_queries"] == 0 and diag["exact_fallback_queries"] == 0, diag
    assert diag["widened_queries"] <= nq // 10, diag


def test_int8_tier_on_hostile_rows(acc, oracle):
    """Rows that quantise badly or not at all: zero rows, rows far outside the fp32 comfort zone
    (no usable norm: e_r = inf, always a candidate), a NaN row, one-hot rows (scale = the whole
    row), heavy-tailed rows (a few huge components: large measured residue), scaled duplicates
    (exact ties), rows equal to a query; plus an allow-mask.  Correctness never depends on the
    quantisation quality — only the number of candidates does."""
    n, d, k = 30000, 256, 40
    rng = np.random.default_rng(77)
    corpus = oracle.synth_rows(32, 0, n, d)
    q = oracle.synth_rows(32, 1 << 40, 9, d)
    corpus[5] = 0.0; corpus[n - 1] = 0.0; corpus[4000:4016] = 0.0     # a whole block of zero rows
    corpus[777] *= np.float32(1e18); corpus[778] *= np.float32(1e-18)
    corpus[779] *= np.float32(3e-21)                                   # norm^2 below 1e-12 in fp64 too: skipped by the reference
    corpus[900, 3] = np.nan; corpus[901, 7] = np.inf
    corpus[1000:1064] = np.eye(64, d, dtype=np.float32) * np.float32(3.0)
    heavy = rng.choice(n, 2000, replace=False)
    corpus[heavy, rng.integers(0, d, 2000)] *= np.float32(40.0)
    corpus[2000:2040] = q[1] * rng.uniform(0.1, 10.0, (40, 1)).astype(np.float32)   # 40 exact ties at 1.0
    corpus[3000] = q[2]; q[3] = corpus[1003]
    rank = rng.permutation(n).astype(np.uint32)
    r = check(acc, oracle, corpus, q, k, tie_rank=rank, expect_path=0, shadow="i8", expect_tier=_lib.TIER_I8)
    assert r.diag["exact_fallback_queries"] == 0
    # allow-mask: masked rows are not part of the scan
    mask = rng.random(n) < 0.6
    got = run(acc, corpus, q, k, -1.0, tie_rank=rank, shadow="i8", mask=mask)
    assert got.diag["filter_tier"] == _lib.TIER_I8
    allowed = np.flatnonzero(mask)
    for qi in range(len(q)):
        rows, sims, _, _ = oracle.scan_cosine(corpus[allowed], q[qi], k, -1.0, rank.astype(np.uint64)[allowed])
        assert np.array_equal(got.rows[qi, :len(rows)], allowed[rows]) and got.counts[qi] == len(rows)
        assert np.array_equal(got.scores[qi, :len(rows)].view(np.uint32), sims.view(np.uint32))


def test_int8_shadow_built_in_appends_equals_one_build(acc, oracle):
    """The rows of a 64-row block share one scale, so an append that starts inside a block must
    re-quantise that block's earlier rows: building in ragged appends == building once."""
    import torch
    n, d = 5003, 320
    corpus = oracle.synth_rows(34, 0, n, d)
    corpus[1001] *= np.float32(7.0)                     # a row with a different largest component
    tc = torch.from_numpy(corpus).cuda()
    nb = (n + 63) // 64
    npad = nb * 64                                       # the shadow is padded to whole 64-row blocks (YAMS_SCAN_I8_SHADOW_ROWS)
    a8 = torch.full((npad, d), 99, dtype=torch.int8, device="cuda"); am = torch.zeros((nb, 2), dtype=torch.float32, device="cuda")
    b8 = torch.full((npad, d), 99, dtype=torch.int8, device="cuda"); bm = torch.zeros((nb, 2), dtype=torch.float32, device="cuda")
    acc.build_shadow_i8_device(tc.data_ptr(), n, d, a8.data_ptr(), am.data_ptr())
    first = 0
    for step in (1000, 1, 7, 2995, 1000):               # 1000 and 1001 fall inside a block
        acc.build_shadow_i8_device(tc.data_ptr(), step, d, b8.data_ptr(), bm.data_ptr(), first_row=first)
        first += step
    assert first == n
    acc.synchronize()
    assert torch.equal(a8, b8) and torch.equal(am, bm)
    # the shadow is stored blocked — [row / 16][slab][16 rows][4 positions][16 B], position p of row r holding
    # chunk p ^ swz(r), swz = (0, 2, 3, 1)[(r >> 2) & 3] (scan_i8_kernel.hip, i8_blocked_offset): undo it
    a8 = unblock_i8_shadow(a8, d)
    assert (a8[n:] == 0).all()                           # padding rows are defined (zero)
    a8 = a8[:n]
    unit = tc.double() / tc.double().norm(dim=-1, keepdim=True)
    sc = am[:, 0].double().repeat_interleave(64)[:n, None]
    assert ((unit - a8.double() * sc).norm(dim=-1) <= am[:, 1].double().repeat_interleave(64)[:n]).all()
    assert (a8.abs().amax(dim=-1).view(-1)[:64 * (n // 64)].view(-1, 64).amax(dim=-1) == 127).all()   # every block uses its full range
    # the rotated layout: ragged appends == one build, too (a row's rotation does not depend on its neighbours)
    a8 = torch.full((npad, d), 99, dtype=torch.int8, device="cuda"); am = torch.zeros((nb, 2), dtype=torch.float32, device="cuda")
    b8 = torch.full((npad, d), 99, dtype=torch.int8, device="cuda"); bm = torch.zeros((nb, 2), dtype=torch.float32, device="cuda")
    acc.build_shadow_i8_device(tc.data_ptr(), n, d, a8.data_ptr(), am.data_ptr(), i8_flags=_lib.I8_ROTATED)
    first = 0
    for step in (1000, 1, 7, 2995, 1000):
        acc.build_shadow_i8_device(tc.data_ptr(), step, d, b8.data_ptr(), bm.data_ptr(), first_row=first, i8_flags=_lib.I8_ROTATED)
        first += step
    acc.synchronize()
    assert torch.equal(a8, b8) and torch.equal(am, bm)
    assert (unblock_i8_shadow(a8, d)[:64 * (n // 64)].abs().amax(dim=-1).view(-1, 64).amax(dim=-1) == 127).all()


def test_int8_tier_widens_escalates_and_falls_back(acc, oracle):
    """The int8 tier's unproven queries take the same road as the bf16 tier's: widen to the whole
    list, then the split filter as a nested batch, then the exhaustive fp64 pass."""
    n, d = 20000, 256
    rng = np.random.default_rng(93)
    corpus = oracle.synth_rows(33, 0, n, d)
    q = oracle.synth_rows(33, 1 << 40, 5, d)
    q[0] *= np.float32(3.0)
    corpus[1::4] = q[0] * np.float32(0.5)                   # 5000 exact ties: list overflow -> exhaustive
    qu = (q[2] / np.linalg.norm(q[2])).astype(np.float64)
    pool = np.setdiff1d(np.arange(n), np.arange(1, n, 4))
    crowd = rng.choice(pool, 3000, replace=False)
    for r_, s_ in zip(crowd, np.linspace(0.990, 0.999, 3000)):  # crowded top: escalates to the split filter
        v = rng.standard_normal(d); v -= (v @ qu) * qu; v /= np.linalg.norm(v)
        corpus[r_] = (s_ * qu + np.sqrt(1.0 - s_ * s_) * v).astype(np.float32) * np.float32(rng.uniform(0.5, 2.0))
    corpus[pool[100:100 + 300]] = q[4]                       # 300 ties: more than k' = 214, fewer than the list: widened
    rank = np.random.default_rng(35).permutation(n).astype(np.uint32)
    import torch
    from yams_amd.accel import Accel
    fresh = Accel(0, torch.cuda.current_stream().cuda_stream)  # (what a context learnt about another corpus at this address would change the road taken)
    try:
        r = check(fresh, oracle, corpus, q, 50, tie_rank=rank, expect_path=0, shadow="i8", expect_tier=_lib.TIER_I8)
    finally:
        fresh.close()
    assert r.diag["escalated_queries"] >= 1 and r.diag["exact_fallback_queries"] >= 1 and r.diag["widened_queries"] >= 1, r.diag


# ---- the plugin vtable door ------------------------------------------------------------------------
def test_vector_scan_vtable(accel_lib, oracle):
    L = accel_lib
    assert L.yams_plugin_init(b'{"device": 0}', None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 1, C.byref(p)) == 0
    vt = C.cast(p, C.POINTER(_lib.VectorScanV1)).contents
    cid = C.c_uint64()
    assert vt.corpus_create(None, 64, C.byref(cid)) == 0
    corpus = oracle.synth_rows(15, 0, 9000, 64)
    half = 4000
    assert vt.corpus_append(None, cid, corpus[:half].ctypes.data_as(_lib.f32p), half) == 0
    assert vt.corpus_append(None, cid, corpus[half:].ctypes.data_as(_lib.f32p), 9000 - half) == 0
    n = C.c_uint64(); dim = C.c_uint32()
    assert vt.corpus_size(None, cid, C.byref(n), C.byref(dim)) == 0 and (n.value, dim.value) == (9000, 64)
    q = oracle.synth_rows(15, 1 << 40, 3, 64)
    hits = C.POINTER(_lib.ScanHit)(); counts = _lib.u32p(); diag = _lib.ScanDiag()
    # dimension mismatch -> InvalidArgument (vector_database.cpp:545-550)
    assert vt.search_batch(None, cid, q.ctypes.data_as(_lib.f32p), 3, 32, 5, -1.0, 0,
                           C.byref(hits), C.byref(counts), None) == _lib.YAMS_ERR_INVALID_ARG
    assert vt.search_batch(None, cid, q.ctypes.data_as(_lib.f32p), 3, 64, 5, -1.0, 0,
                           C.byref(hits), C.byref(counts), C.byref(diag)) == 0
    for qi in range(3):
        rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], 5)
        assert counts[qi] == 5
        assert [hits[qi * 5 + i].row for i in range(5)] == list(rows)
        assert [hits[qi * 5 + i].similarity for i in range(5)] == [float(x) for x in sims]
        assert abs(hits[qi * 5].distance - (1.0 - hits[qi * 5].similarity)) < 1e-7
    vt.free_hits(None, hits, counts)
    assert vt.search_batch(None, C.c_uint64(999), q.ctypes.data_as(_lib.f32p), 3, 64, 5, -1.0, 0,
                           C.byref(hits), C.byref(counts), None) == _lib.YAMS_ERR_NOT_FOUND
    info = C.c_void_p()
    assert vt.get_runtime_info_json(None, C.byref(info)) == 0 and b"gfx950" in C.string_at(info)
    vt.free_string(None, info)
    assert vt.corpus_destroy(None, cid) == 0
    L.yams_plugin_shutdown()


def _vt(L, config):
    L.yams_plugin_shutdown()                  # (a test that failed half-way leaves the plugin initialised)
    assert L.yams_plugin_init(config, None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 1, C.byref(p)) == 0
    return C.cast(p, C.POINTER(_lib.VectorScanV1)).contents


def _vt_search(vt, cid, q, k, thr=-1.0, metric=0, mask=None, flags=0):
    nq, d = q.shape
    hits = C.POINTER(_lib.ScanHit)(); counts = _lib.u32p(); diag = _lib.ScanDiag()
    mp = mask.ctypes.data_as(_lib.u32p) if mask is not None else None
    st = vt.search_batch_ex(None, cid, q.ctypes.data_as(_lib.f32p), nq, d, k, thr, metric, flags, mp,
                            C.byref(hits), C.byref(counts), C.byref(diag))
    assert st == 0, st
    rows = [[hits[qi * k + i].row for i in range(counts[qi])] for qi in range(nq)]
    sims = [np.array([hits[qi * k + i].similarity for i in range(counts[qi])], np.float32) for qi in range(nq)]
    vt.free_hits(None, hits, counts)
    return rows, sims, diag.as_dict()


@pytest.mark.parametrize("config", [b'{"device": 0}', b'{"devices": [0, 0, 0], "stripe_rows": 4096, "search_slots": 2}'])
def test_plugin_mirror_grows_in_place_and_is_dealt_to_shards(accel_lib, oracle, config):
    """vector_scan_v1 with the corpus on one shard or dealt to three (contexts on one device, stripes of
    4096 rows): ragged appends that straddle stripes (the mirror grows in place, nothing is re-uploaded),
    a corpus-wide chunk_id ranking, an allow-mask, batches on both sides of the int8 tier's threshold —
    every answer equals the oracle over the whole corpus."""
    L = accel_lib
    vt = _vt(L, config)
    n, d, k = 70_001, 256, 20
    corpus = oracle.synth_rows(51, 0, n, d)
    corpus[100] = corpus[9000] = corpus[50_000]                 # exact ties on different shards
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    pos = 0
    for step in (1, 4095, 4096, 10_000, 3, 40_000, n):          # ragged appends
        step = min(step, n - pos)
        if step == 0:
            break
        part = np.ascontiguousarray(corpus[pos:pos + step])
        assert vt.corpus_append(None, cid, part.ctypes.data_as(_lib.f32p), step) == 0
        pos += step
    nn = C.c_uint64(); dd = C.c_uint32()
    assert vt.corpus_size(None, cid, C.byref(nn), C.byref(dd)) == 0 and (nn.value, dd.value) == (n, d)
    q = oracle.synth_rows(51, 1 << 40, 140, d)
    q[0] = corpus[100]
    rank = np.random.default_rng(8).permutation(n).astype(np.uint32)
    for use_rank in (False, True):
        if use_rank:
            assert vt.corpus_set_tie_ranks(None, cid, rank.ctypes.data_as(_lib.u32p), n) == 0
        tr = rank.astype(np.uint64) if use_rank else None
        for nq in (3, 140):                                      # narrow bf16 form / int8 tier
            rows, sims, diag = _vt_search(vt, cid, np.ascontiguousarray(q[:nq]), k)
            assert diag["filter_tier"] == (_lib.TIER_I8 if nq > 128 else _lib.TIER_BF16), diag
            for qi in list(range(nq))[:6]:
                orow, osim, _, _ = oracle.scan_cosine(corpus, q[qi], k, -1.0, tr)
                assert rows[qi] == list(orow), (config, use_rank, nq, qi, rows[qi][:6], orow[:6])
                assert np.array_equal(sims[qi].view(np.uint32), osim.view(np.uint32))
    # allow-mask over global rows (document_hash / candidate_hashes restriction)
    allow = np.random.default_rng(9).random(n) < 0.5
    bits = np.zeros((n + 31) // 32 * 32, np.uint8); bits[:n] = allow
    words = np.ascontiguousarray(np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel())
    rows, sims, _ = _vt_search(vt, cid, np.ascontiguousarray(q[:4]), k, mask=words)
    idx = np.flatnonzero(allow)
    for qi in range(4):
        orow, osim, _, _ = oracle.scan_cosine(corpus[idx], q[qi], k, -1.0, rank.astype(np.uint64)[idx])
        assert rows[qi] == list(idx[orow]) and np.array_equal(sims[qi].view(np.uint32), osim.view(np.uint32))
    # L2 (vec0 semantics) through the same mirror
    rows, _, _ = _vt_search(vt, cid, np.ascontiguousarray(q[:3]), k, thr=-1.0, metric=1)
    for qi in range(3):
        orow, _, _ = oracle.scan_l2(corpus, q[qi], k, -1.0, rank.astype(np.uint64))
        assert rows[qi] == list(orow)
    # ... and a batch on the int8 side of the threshold: the mirror's shadows carry L2 on that tier too
    rows, _, diag = _vt_search(vt, cid, np.ascontiguousarray(q), k, thr=-1.0, metric=1)
    assert diag["filter_tier"] == _lib.TIER_I8, diag
    for qi in (0, 1, 70, 139):
        orow, _, _ = oracle.scan_l2(corpus, q[qi], k, -1.0, rank.astype(np.uint64))
        assert rows[qi] == list(orow), qi
    assert vt.corpus_clear(None, cid) == 0 and vt.corpus_size(None, cid, C.byref(nn), None) == 0 and nn.value == 0
    assert vt.corpus_append(None, cid, corpus[:5000].ctypes.data_as(_lib.f32p), 5000) == 0    # reusable after clear
    rows, _, _ = _vt_search(vt, cid, np.ascontiguousarray(q[1:2]), 5)
    assert rows[0] == list(oracle.scan_cosine(corpus[:5000], q[1], 5, -1.0)[0])
    assert vt.corpus_destroy(None, cid) == 0
    L.yams_plugin_shutdown()


def test_plugin_serves_concurrent_calls(accel_lib, oracle):
    """The reference serves searches under a shared lock (vector_database.cpp:539,618).  Here every call
    leases its own contexts: 8 threads searching + 1 thread chunking finish sooner than the same calls
    one after the other, and every result is still exact."""
    import threading, time
    L = accel_lib
    vt = _vt(L, b'{"device": 0, "search_slots": 4}')
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"chunker_v1", 1, C.byref(p)) == 0
    ck = C.cast(p, C.POINTER(_lib.ChunkerV1)).contents
    n, d, k, nq = 400_000, 256, 10, 8
    corpus = oracle.synth_rows(52, 0, n, d)
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    assert vt.corpus_append(None, cid, corpus.ctypes.data_as(_lib.f32p), n) == 0
    qs = [oracle.synth_rows(52, (1 << 40) + 100 * t, nq, d) for t in range(8)]
    blob = np.random.default_rng(3).integers(0, 256, 48 << 20, dtype=np.uint8)
    cfg = _lib.CdcConfig(); ck.get_default_config(None, _lib.CDC_STREAMING, C.byref(cfg))
    out = {}

    def search(t, reps=6):
        for _ in range(reps):
            out[t] = _vt_search(vt, cid, qs[t], k)[0]

    def chunk():
        chunks = C.POINTER(_lib.ChunkRef)(); cnt = C.c_size_t()
        assert ck.chunk_data(None, blob.ctypes.data_as(_lib.u8p), blob.size, C.byref(cfg), C.byref(chunks), C.byref(cnt)) == 0
        out["chunks"] = cnt.value
        ck.free_chunks(None, chunks, cnt)

    search(0, 1); chunk()                                        # warm-up (workspaces)
    t0 = time.perf_counter()
    for t in range(8):
        search(t)
    chunk()
    serial = time.perf_counter() - t0
    th = [threading.Thread(target=search, args=(t,)) for t in range(8)] + [threading.Thread(target=chunk)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    conc = time.perf_counter() - t0
    for t in range(8):
        for qi in range(2):
            assert out[t][qi] == list(oracle.scan_cosine(corpus, qs[t][qi], k, -1.0)[0])
    assert out["chunks"] > 1000
    assert conc < 0.9 * serial, (conc, serial)
    assert vt.corpus_destroy(None, cid) == 0
    L.yams_plugin_shutdown()


# ---- randomized differential sweep ------------------------------------------------------------------
def test_randomized_differential_sweep(acc, oracle):
    """48 seeded random configurations (shape, k, metric, threshold, tie ranks, allow-mask, filter
    tier, shadow on/off, record path) — every one must match the oracle bit for bit."""
    rng = np.random.default_rng(20260925)
    dims = [4, 8, 20, 32, 48, 64, 96, 100, 128, 256, 384]
    for case in range(48):
        d = int(rng.choice(dims))
        n = int(rng.choice([1, 3, 70, 900, 4096, 5000, 20000, 66000]))
        nq = int(rng.integers(1, 24))
        k = int(rng.choice([1, 2, 10, 37, 100, 300]))
        metric = SCAN_L2 if rng.random() < 0.3 else SCAN_COSINE
        thr = float(rng.choice([-1.0, 0.0, 0.05, 0.3]))
        corpus = oracle.synth_rows(1000 + case, 0, n, d)
        q = oracle.synth_rows(1000 + case, 1 << 40, nq, d)
        if rng.random() < 0.5:                                   # scaled rows, duplicates, degenerate rows
            corpus *= rng.uniform(1e-3, 1e3, (n, 1)).astype(np.float32)
            if n > 10:
                corpus[rng.integers(0, n, 5)] = corpus[rng.integers(0, n)]
                corpus[rng.integers(0, n)] = 0
                corpus[rng.integers(0, n)] *= np.float32(1e-6)
        rank = rng.permutation(n).astype(np.uint32) if rng.random() < 0.5 else None
        allowed = None
        if rng.random() < 0.4 and n > 3:
            allowed = np.sort(rng.choice(n, int(rng.integers(1, n)), replace=False))
        flags = int(rng.choice([0, 0, FLAG_SPLIT_FILTER, FLAG_F32_FILTER]))
        record = metric == SCAN_COSINE and rng.random() < 0.3
        if record:
            flags |= FLAG_RECORD_PATH
        shadow = bool(rng.random() < 0.7)
        # ---- device
        dc = acc.to_device(corpus)
        db = dn = dm = dr = di = None
        if shadow and d % 4 == 0:
            db, dn = acc.alloc(corpus.size * 2), acc.alloc(n * 4)
            acc.build_shadow_device(dc.ptr, n, d, db.ptr, dn.ptr)
        if rank is not None:
            inv = np.empty_like(rank); inv[rank] = np.arange(n, dtype=rank.dtype)
            dr, di = acc.to_device(rank), acc.to_device(inv)
        n_allowed = 0
        if allowed is not None:
            bits = np.zeros((n + 31) // 32 * 32, np.uint8); bits[allowed] = 1
            words = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
            dm = acc.to_device(words); n_allowed = len(allowed)
        view = acc.corpus_view(dc.ptr, n, d, dr.ptr if dr else None, di.ptr if di else None, 0,
                               dm.ptr if dm else None, n_allowed,
                               rows_bf16_ptr=db.ptr if db else None, rows_nsq_ptr=dn.ptr if dn else None)
        r = acc.scan_topk(view, q, k, thr, metric, flags)
        # ---- oracle
        tr64 = None if rank is None else rank.astype(np.uint64)
        tag = (case, n, d, nq, k, metric, thr, flags, shadow, allowed is not None, r.diag)
        for qi in range(nq):
            if record:
                allow8 = None
                if allowed is not None:
                    allow8 = np.zeros(n, np.uint8); allow8[allowed] = 1
                rows, sims, _ = oracle.scan_cosine_records(corpus, q[qi], k, thr, tr64, allow8)
                dist = None
            else:
                sub = corpus if allowed is None else corpus[allowed]
                sr = tr64 if (allowed is None or tr64 is None) else tr64[allowed]
                if metric == SCAN_COSINE:
                    rows, sims, _, _ = oracle.scan_cosine(sub, q[qi], k, thr, sr); dist = None
                else:
                    rows, dist, sims = oracle.scan_l2(sub, q[qi], k, thr, sr)
                if allowed is not None:
                    rows = np.asarray(allowed)[rows]
            cnt = int(r.counts[qi])
            assert cnt == len(rows), (qi, cnt, len(rows), tag)
            assert np.array_equal(r.rows[qi, :cnt], rows), (qi, tag)
            assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), (qi, tag)
            if dist is not None:
                assert np.array_equal(r.dist[qi, :cnt].view(np.uint32), dist.view(np.uint32)), (qi, tag)


def test_clustered_corpus_stays_exact(acc, oracle):
    """A near-duplicate-heavy corpus (40 tight clusters of ~3000 rows, queries at cluster centres): thousands of
    rows within the single-pass bound of every top-k boundary.  Whatever mix of widening, escalation
    and exhaustive scoring the tiers choose, the answer is the oracle's, bit for bit."""
    rng = np.random.default_rng(404)
    n, d, nc = 120000, 64, 40
    centres = rng.standard_normal((nc, d)).astype(np.float32)
    assign = rng.integers(0, nc, n)
    corpus = (centres[assign] + 0.02 * rng.standard_normal((n, d))).astype(np.float32)
    corpus *= rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    q = (centres[:6] + 0.01 * rng.standard_normal((6, d))).astype(np.float32)
    rank = rng.permutation(n).astype(np.uint32)
    r = check(acc, oracle, corpus, q, 100, metric=SCAN_COSINE, tie_rank=rank, expect_path=0)
    # ~3000 cluster members within 1e-4 of each other: neither filter tier can prove a top-100
    assert r.diag["widened_queries"] >= 6 and r.diag["escalated_queries"] == 6 and r.diag["exact_fallback_queries"] == 6, r.diag
    # L2: the random row scales spread the distances, the filter separates them easily
    check(acc, oracle, corpus, q, 100, metric=SCAN_L2, tie_rank=rank, expect_path=0)


def test_batches_above_4096_queries_run_as_slices(acc, oracle):
    """yams_scan_topk_device slices batches of more than 4096 queries; results, counts and the
    summed diagnostics are those of one batch; an invalid query anywhere still fails the call."""
    n, d, k = 9000, 64, 7
    corpus = oracle.synth_rows(31, 0, n, d)
    q = oracle.synth_rows(31, 1 << 40, 4096 + 333, d)
    r = run(acc, corpus, q, k)
    assert r.diag["rows_visited"] == q.shape[0] * n and r.diag["returned_rows"] == q.shape[0] * k
    for qi in (0, 4095, 4096, 4097, q.shape[0] - 1):
        rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, -1.0)
        assert np.array_equal(r.rows[qi], rows) and np.array_equal(r.scores[qi].view(np.uint32), sims.view(np.uint32))
    a = run(acc, corpus, q[4000:4200], k)          # the same queries inside one slice
    assert np.array_equal(a.rows, r.rows[4000:4200]) and np.array_equal(a.scores.view(np.uint32), r.scores[4000:4200].view(np.uint32))
    q[4200] = 0.0
    with pytest.raises(_lib.AccelError) as e:
        run(acc, corpus, q, k)
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG


def test_randomised_self_consistency_of_all_scan_paths():
    """tests/stress_scan.py: 150 random shapes (rows, dim, batch 1..200, k, metric, threshold,
    allow-mask, clustered tops, zero / huge rows); the default path (narrow or 256-query filter,
    widening, escalation, fallback) must equal the exhaustive fp64 path and the wide form bit for
    bit.  Round 6: rows of four component distributions, either int8 shadow layout, every int8 call twice (learnt hints).
    (1680 further cases were run by hand with seeds 1-5 in round 4, 900 in round 6: no mismatch.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_scan.py"), "--cases", "150", "--seed", "7"],
                       capture_output=True, text=True, timeout=280)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1])
    assert res["cases"] == 150 and res["mismatches"] == 0, res
    assert any("widened1" in p for p in res["paths"]) and any(p.startswith("path1") for p in res["paths"]), res


def test_every_rescored_candidate_lies_inside_its_filter_bound():
    """Bound honesty (round 6; the measurement build counts it in rescore_select_kernel): for EVERY candidate the tiers
    re-score, the filter's score against the exact similarity — |score - cos| <= the tier's error bound on the f32 / bf16 /
    split tiers, cos <= score on the int8 tier (its score is an upper bound), in either shadow layout; under L2, on every tier,
    g = q.x - |x|^2 / 2 <= score.  The proofs stand on
    exactly this; a final result can be right while a bound is not.  120 random cases of tests/stress_scan.py (uniform /
    Gaussian / power-law / outlier-dimension rows, masks, clusters, learnt hints) on the measurement build: no candidate
    outside its bound, no mismatch.  (Round 6 ran 1 950 cases, 4 000 calls, by hand: none.)"""
    import json, os, re, subprocess, sys
    from yams_amd import build as _build
    if not os.path.exists(_build.MEASURE_LIB):
        _build.build(measure=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, YAMS_ACCEL_MEASURE_LIB="1", YAMS_ACCEL_DUMP_NEEDED="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_scan.py"), "--cases", "120", "--seed", "21"],
                       capture_output=True, text=True, timeout=280, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1])
    assert res["cases"] == 120 and res["mismatches"] == 0, res
    counts = [(int(m.group(1)), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"bound honesty: (\d+) of (\d+) re-scored candidates outside their filter bound \(tier (\d)\)", r.stderr)]
    assert len(counts) >= 150 and {t for _, _, t in counts} >= {1, 2}, (len(counts), {t for _, _, t in counts})
    assert all(v == 0 for v, _, _ in counts), [c for c in counts if c[0]][:5]


def test_soak_of_the_bench_configuration_is_deterministic():
    """tests/soak_scan.py: the bench's shape (12.5M x 768 shard, 1024 queries, top-100, two lanes behind one sweep gate),
    400 batches over four rotating query batches — every result bit-identical to the first result of its query batch
    (races between lanes, in the persistent sweep's strip counters and pacing, in the candidate lists would show here;
    1500 + 800 batches were run by hand in round 4: no difference)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "soak_scan.py"), "--batches", "400"],
                       capture_output=True, text=True, timeout=280)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1])
    assert res["batches"] == 400 and res["differences"] == 0, res


def test_sweep_hold_keeps_the_gate_closed_until_the_collective_is_enqueued(oracle):
    """yams_accel_ctx_set_sweep_hold (the one-process-per-GPU form of the exchange fence): with the hold on, the
    gate stays closed behind a context's sweep — a second context's scan blocks at its own sweep until the first calls
    release_sweep_hold (after enqueuing its collective) — and nothing deadlocks: releasing without a hold is a no-op, a
    destroyed context gives the gate back, results stay oracle-exact."""
    import threading, time
    from yams_amd.accel import Accel, SweepGate
    n, d, nq, k = 60_000, 256, 140, 10
    corpus = oracle.synth_rows(91, 0, n, d); q = oracle.synth_rows(91, 1 << 40, nq, d)
    a, b = Accel(0), Accel(0)
    gate = SweepGate(0)
    a.set_gate(gate); b.set_gate(gate)
    dc = a.to_device(corpus)
    d8, dm8 = a.alloc(_lib.i8_shadow_rows(n) * d), a.alloc((n + 15) // 16 * 8)
    a.build_shadow_i8_device(dc.ptr, n, d, d8.ptr, dm8.ptr); a.synchronize()
    view = a.corpus_view(dc.ptr, n, d, rows_i8_ptr=d8.ptr, rows_i8_meta_ptr=dm8.ptr)
    a.release_sweep_hold()                       # no hold: a no-op
    a.set_sweep_hold(True)
    ra = a.scan_topk(view, q, k, -1.0)           # sweeps, then keeps the gate closed
    assert ra.diag["filter_tier"] == _lib.TIER_I8
    done = {}
    t = threading.Thread(target=lambda: done.setdefault("r", b.scan_topk(view, q, k, -1.0)))
    t.start()
    time.sleep(0.3)
    assert "r" not in done                       # b waits at the gate
    a.release_sweep_hold()                       # (the collective would have been enqueued on a side stream here)
    t.join(timeout=30)
    assert "r" in done
    for r in (ra, done["r"]):
        for qi in (0, 70, 139):
            rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, -1.0)
            assert np.array_equal(r.rows[qi], rows) and np.array_equal(r.scores[qi].view(np.uint32), sims.view(np.uint32))
    # a context destroyed while holding gives the gate back
    a.scan_topk(view, q, k, -1.0)
    t = threading.Thread(target=lambda: done.setdefault("r2", b.scan_topk(view, q, k, -1.0)))
    t.start(); time.sleep(0.2)
    assert "r2" not in done
    a.set_sweep_hold(False)                      # turning the hold off releases it
    t.join(timeout=30)
    assert "r2" in done
    a.close(); b.close(); gate.close()


# ---- the reference's own loop, compiled (oracle/_ref/libyams_scan_ref.so), as golden vectors ------------------------------
@pytest.mark.parametrize("shadow", [True, False, "both"])
def test_hip_scan_reproduces_the_reference_compiled_golden_vectors(acc, oracle, shadow):
    """tests/golden/scan.json holds what SqliteVecBackend::Impl::bruteForceSearchUnlocked ITSELF returned (both paths;
    compiled from /root/reference by oracle/Makefile, recorded by tests/golden/make_scan_golden.py): the HIP path must
    return the same rows in the same order with the same score bits — config 1 on the reference's mt19937 recipe,
    Philox rows at dim 768 / 384 top-100, k > n, ties under shuffled chunk ids, skipped rows and +-FLT_MAX/4, invalid
    queries, the metadata-filter path (TopK and AllMatching)."""
    with open(os.path.join(_cases.GOLDEN, "scan.json")) as f:
        g = json.load(f)
    n_cases = 0
    for case in g["cases"]:
        corpus, queries, tie_rank, allow = _cases.golden_scan_inputs(oracle, case)
        if shadow == "both" and not (corpus.shape[1] % 64 == 0 and corpus.shape[1] >= 256):
            continue
        flags, mask, k = 0, None, case["k"]
        if case["path"] == "fast" and allow is not None:
            mask = allow.astype(bool)                    # candidate_hashes: the host folds the restriction into the allow-mask
        if case["path"] == "record":
            flags, mask = FLAG_RECORD_PATH, allow.astype(bool)
            if case.get("all_matching"):
                k = int(allow.sum())                     # ExactRowSelection::AllMatching = every allowed row (:4398-4400)
        good = [qi for qi, e in enumerate(case["expected"]) if not e.get("error")]
        for qi, e in enumerate(case["expected"]):
            if e.get("error"):                           # :4127-4130 InvalidArgument; a batch fails as a whole (:1635-1647)
                with pytest.raises(_lib.AccelError) as err:
                    run(acc, corpus, queries[qi:qi + 1], k, case["threshold"], SCAN_COSINE, flags, tie_rank, shadow=shadow, mask=mask)
                assert err.value.status == _lib.YAMS_ERR_INVALID_ARG
        if not good:
            continue
        r = run(acc, corpus, queries[good], k, case["threshold"], SCAN_COSINE, flags, tie_rank, shadow=shadow, mask=mask)
        for j, qi in enumerate(good):
            e = case["expected"][qi]
            cnt = int(r.counts[j])
            assert cnt == len(e["rows"]), (case["name"], qi, cnt, len(e["rows"]), r.diag)
            assert r.rows[j, :cnt].tolist() == e["rows"], (case["name"], qi, r.diag)
            assert [int(x) for x in r.scores[j, :cnt].view(np.uint32)] == e["score_bits"], (case["name"], qi)
        n_cases += 1
    assert n_cases >= (6 if shadow == "both" else 15)


# ---- the reference's own vec0SearchUnlocked, compiled, as golden vectors (round 6) ---------------------------------------------
L2_GOLDEN_FLAGS = {"f64": _lib.FLAG_L2_ACC_F64, "f32": _lib.FLAG_L2_ACC_F32, "f32x8": _lib.FLAG_L2_ACC_F32X8, "f32x16": _lib.FLAG_L2_ACC_F32X16,
                   "f32_fma": _lib.FLAG_L2_ACC_F32 | _lib.FLAG_L2_ACC_FUSED, "f32x8_fma": _lib.FLAG_L2_ACC_F32X8 | _lib.FLAG_L2_ACC_FUSED,
                   "f32x16_fma": _lib.FLAG_L2_ACC_F32X16 | _lib.FLAG_L2_ACC_FUSED}


@pytest.mark.parametrize("shadow", [True, "both"])
def test_hip_l2_scan_reproduces_the_reference_compiled_vec0_golden_vectors(acc, oracle, shadow):
    """tests/golden/scan_l2.json holds what SqliteVecBackend::Impl::vec0SearchUnlocked ITSELF returned (compiled from
    /root/reference, over the harness's vec0 module with each of the seven distance definitions plugged in): the HIP path under
    the matching accumulate flag must return the same rows in the same order with the same cosine bits — k nearest then the
    threshold, k > n, EQUAL DISTANCES IN ROWID ORDER although the view carries a shuffled chunk_id ranking (it belongs to the
    cosine comparator), the candidate restriction as an allow-mask, and the case whose answer depends on the definition."""
    with open(os.path.join(_cases.GOLDEN, "scan_l2.json")) as f:
        g = json.load(f)
    n_checked = 0
    for case in g["cases"]:
        corpus, queries, tie_rank, allow = _cases.golden_scan_inputs(oracle, case)
        if shadow == "both" and not (corpus.shape[1] % 64 == 0 and corpus.shape[1] >= 256):
            continue
        mask = allow.astype(bool) if allow is not None else None
        for name, fl in L2_GOLDEN_FLAGS.items():
            if name in case["same_as_f64"] and name not in ("f32x8_fma", "f32"):
                continue        # (identical expectations: two of the fp32 forms stand for the others)
            exp = case["expected"]["f64" if name in case["same_as_f64"] else name]
            r = run(acc, corpus, queries, case["k"], case["threshold"], SCAN_L2, fl, tie_rank, shadow=shadow, mask=mask)
            for qi, e in enumerate(exp):
                cnt = int(r.counts[qi])
                assert cnt == len(e["rows"]), (case["name"], name, qi, cnt, len(e["rows"]), r.diag)
                assert r.rows[qi, :cnt].tolist() == e["rows"], (case["name"], name, qi, r.diag)
                assert [int(x) for x in r.scores[qi, :cnt].view(np.uint32)] == e["score_bits"], (case["name"], name, qi)
                n_checked += 1
    assert n_checked >= (60 if shadow == "both" else 150)


# ---- corpora that are not uniform on the sphere (round 6) --------------------------------------------------------------------
def _clustered(n, d, n_clusters, seed, nq, spread=0.35):
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((n_clusters, d)).astype(np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    sigma = np.float32(spread / np.sqrt(d))
    x = centres[rng.integers(0, n_clusters, n)] + sigma * rng.standard_normal((n, d)).astype(np.float32)
    q = centres[rng.integers(0, n_clusters, nq)] + sigma * rng.standard_normal((nq, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32), (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)


def test_clustered_corpus_is_proven_by_the_int8_retry(acc, oracle):
    """Queries near the centre of a cluster of ~1000 rows whose similarities differ by less than the int8 bound is wide: the
    sampled threshold sits inside the cloud and every proof of stage 1 fails.  Stage 2a filters those queries again on the
    int8 tier with the threshold the proof asks for — the k-th best exact score found, one ulp down — and proves them without
    the split-bf16 sweep; rows, order and score bits equal the oracle's."""
    import torch
    from yams_amd.accel import Accel
    corpus, q = _clustered(300_000, 256, 230, 71, 200)         # ~1300 rows per cluster: more than the sampled threshold lists
    fresh = Accel(0, torch.cuda.current_stream().cuda_stream)  # (a context that has learnt nothing about any corpus at this address)
    try:
        r = check(fresh, oracle, corpus, q, 100, max_queries=24, expect_path=0, shadow="i8", expect_tier=_lib.TIER_I8)
    finally:
        fresh.close()
    assert r.diag["retried_queries"] >= 10, r.diag
    assert r.diag["escalated_queries"] <= r.diag["retried_queries"] // 3 and r.diag["exact_fallback_queries"] == 0, r.diag
    # the same through the resident-query form and with a threshold (a context that has just served this corpus at this address
    # may already plan deeper lists for it — the depth hint — and need no second pass)
    r2 = check(acc, oracle, corpus, q, 50, thr=0.3, max_queries=12, expect_path=0, shadow="i8", flags=_lib.FLAG_RESIDENT_QUERIES)
    assert r2.diag["retried_queries"] > 0 or r2.diag["widened_queries"] == 0, r2.diag
    assert r2.diag["exact_fallback_queries"] == 0, r2.diag


def test_tight_clusters_are_listed_whole_in_the_first_pass(acc, oracle):
    """The proof-aware threshold (tau_select_kernel, round 6): 1430 clusters of ~1050 rows whose similarities to a query of
    their own cluster lie within a fraction of the int8 bound's width.  For the queries whose cluster has 16 or more rows in
    the sample (six in ten) the rank rule's threshold — the 16th best sampled bound — sits inside the cluster and no list it
    produces can prove a top-100; the sample shows the crowd (best sampled bound within E of that threshold), so the
    threshold drops 2 E below the 6th best sampled bound — under the whole cluster — and the FIRST int8 pass lists everything
    the proof needs: a second pass only for the few queries whose cluster the sample over-represents (more than 22 rows: the
    deeper list might not fit), no escalation to speak of; rows, order and score bits are the oracle's."""
    nq = 160
    corpus, q = _clustered(1_500_000, 256, 1430, 73, nq, spread=0.15)
    r = check(acc, oracle, corpus, q, 100, max_queries=8, expect_path=0, shadow="i8", expect_tier=_lib.TIER_I8)
    assert r.diag["retried_queries"] <= nq // 5 and r.diag["escalated_queries"] <= 5 and r.diag["exact_fallback_queries"] == 0, r.diag
    assert r.diag["rescored_rows"] >= nq * 900, r.diag      # the lists hold the clusters


def test_anisotropic_corpus_teaches_the_context_to_start_on_the_bf16_tier(acc, oracle):
    """Rows with a power-law spectrum (a few large components, a long tail): the int8 shadow's residue — hence its bound — is
    several times the isotropic one, every query of an int8 batch escalates.  The context remembers that per corpus: the
    next batches of more than 128 queries start on the bf16 tier (no escalation), results identical and oracle-exact."""
    import torch
    rng = np.random.default_rng(72)
    n, d, nq, k = 200_000, 256, 160, 50
    scale = (np.arange(1, d + 1, dtype=np.float32) ** -0.5)
    x = (rng.standard_normal((n, d)).astype(np.float32) * scale); x /= np.linalg.norm(x, axis=1, keepdims=True)
    qs = (rng.standard_normal((nq, d)).astype(np.float32) * scale); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    from yams_amd.accel import Accel
    a2 = Accel(0, torch.cuda.current_stream().cuda_stream)       # a context of its own: what it learns must not leak into other tests
    try:
        tc = torch.from_numpy(x).cuda()
        tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
        a2.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
        t8 = torch.empty((_lib.i8_shadow_rows(n), d), dtype=torch.int8, device="cuda"); tm = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
        a2.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm.data_ptr())
        v = a2.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(), rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm.data_ptr())
        first = a2.scan_topk(v, qs, k, -1.0)
        assert first.diag["filter_tier"] == _lib.TIER_I8, first.diag
        later = a2.scan_topk(v, qs, k, -1.0)
        if first.diag["escalated_queries"] * 2 > nq:             # the lesson: the int8 batch escalated, the next one starts on bf16
            assert later.diag["filter_tier"] == 2 and later.diag["escalated_queries"] <= first.diag["escalated_queries"] // 4, (first.diag, later.diag)
        assert np.array_equal(first.rows, later.rows) and np.array_equal(first.scores.view(np.uint32), later.scores.view(np.uint32))
        for qi in range(0, nq, 13):
            rows, sims, _, _ = oracle.scan_cosine(x, qs[qi], k, -1.0)
            assert np.array_equal(later.rows[qi], rows) and np.array_equal(later.scores[qi].view(np.uint32), sims.view(np.uint32)), qi
        small = a2.scan_topk(v, qs[:40], k, -1.0)                  # batches of <= 128 queries keep the library's usual choice
        assert np.array_equal(small.rows, later.rows[:40])
    finally:
        a2.close()


# ---- the rotated int8 layout (round 6) ---------------------------------------------------------------------------------
def _rot_sign_np(idx, salt):
    """scan_i8_kernel.hip rot_sign: +1 / -1 per component index."""
    v = (idx.astype(np.uint64) * 2654435761 + salt * 0x9E3779B9) & 0xffffffff
    v ^= v >> 15; v = (v * 2246822519) & 0xffffffff
    v ^= v >> 13; v = (v * 3266489917) & 0xffffffff
    v ^= v >> 16
    return np.where(v & 1, -1.0, 1.0)


def _rotate_np(v):
    """The map the rotated layout claims, in fp64: R = H_B S_2 H_A S_1 (windows [0, P) and [d - P, d), P = 2^floor(log2 d))."""
    from scipy.linalg import hadamard
    d = v.shape[1]
    p = 1 << (d.bit_length() - 1)
    h = hadamard(p).astype(np.float64) / np.sqrt(p)
    idx = np.arange(d)
    w = v.astype(np.float64) * _rot_sign_np(idx, 1)
    w[:, :p] = w[:, :p] @ h
    if p < d:
        w[:, d - p:] = (w[:, d - p:] * _rot_sign_np(idx[d - p:], 2)) @ h
    return w


def _mixed_rows(rng, n, d):
    """Rows of four kinds: uniform components, a power-law spectrum, a few outlier dimensions, a common mean."""
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x[n // 4:n // 2] = rng.standard_normal((n // 2 - n // 4, d)).astype(np.float32) * (np.arange(1, d + 1, dtype=np.float32) ** -0.5)
    x[n // 2:3 * n // 4] = rng.standard_normal((3 * n // 4 - n // 2, d)).astype(np.float32)
    x[n // 2:3 * n // 4, [3, d // 3, d - 5]] *= np.float32(12.0)
    x[3 * n // 4:] = (0.6 + rng.standard_normal((n - 3 * n // 4, d))).astype(np.float32)
    return x


@pytest.mark.parametrize("d", [256, 384, 768, 1024, 1536, 4096])
def test_rotated_int8_shadow_is_the_rotation_it_claims(acc, d):
    """The residue the rotated shadow RECORDS (meta e_b) bounds the distance between every de-quantised int8 row and the EXACT
    rotation (fp64, scipy's Hadamard matrix) of the exactly normalised row — the property the filter's bound stands on — and
    is not needlessly loose; a dimension that is a power of two takes one transform, the others two overlapping ones."""
    import torch
    rng = np.random.default_rng(600 + d)
    n = 200
    x = _mixed_rows(rng, n, d)
    x[17] = 0.0; x[44, 5] = np.inf
    tc = torch.from_numpy(x).cuda()
    t8 = torch.empty((_lib.i8_shadow_rows(n), d), dtype=torch.int8, device="cuda"); tm = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
    mean = acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm.data_ptr(), want_mean_err=True, i8_flags=_lib.I8_ROTATED)
    xi = unblock_i8_shadow(t8, d).cpu().numpy().astype(np.float64)
    meta = tm.cpu().numpy().astype(np.float64)
    ok = np.isfinite(x).all(axis=1) & (np.abs(x).max(axis=1) > 0)
    unit = np.zeros((n, d)); unit[ok] = x[ok].astype(np.float64) / np.linalg.norm(x[ok].astype(np.float64), axis=1, keepdims=True)
    y = _rotate_np(unit)
    assert np.abs(np.linalg.norm(y[ok], axis=1) - 1.0).max() < 1e-12          # (the reference map is orthogonal)
    worst = 0.0
    for b in range((n + 63) // 64):
        rows = slice(64 * b, min(n, 64 * b + 64))
        dist = np.linalg.norm(meta[b, 0] * xi[rows] - y[rows], axis=1)
        okb = ok[rows]
        assert (xi[rows][~okb] == 0).all()
        assert dist[okb].max() <= meta[b, 1] + 1.75e-6, (b, dist[okb].max(), meta[b, 1])   # + the rotation's own rounding (in the query slop)
        assert meta[b, 1] <= dist[okb].max() * 1.02 + 2e-4, (b, dist[okb].max(), meta[b, 1])
        worst = max(worst, dist[okb].max())
    assert 0 < mean <= worst * 1.02 + 2e-4
    assert (xi[n:] == 0).all()


def test_layout_choice_follows_the_measured_residues(acc):
    """yams_scan_choose_i8_layout_device: uniform components keep the plain layout (the rotation would more than double their
    residue), isotropic Gaussian rows too (nothing to gain), a power-law spectrum and outlier dimensions take the rotated one;
    the two means it reports are those of full builds."""
    import torch
    rng = np.random.default_rng(611)
    n, d = 40_000, 768
    kinds = {"uniform": rng.uniform(-1, 1, (n, d)).astype(np.float32),
             "gauss": rng.standard_normal((n, d)).astype(np.float32),
             "powerlaw": rng.standard_normal((n, d)).astype(np.float32) * (np.arange(1, d + 1, dtype=np.float32) ** -0.5)}
    o = rng.standard_normal((n, d)).astype(np.float32); o[:, [7, 300, 301, 700]] *= np.float32(10.0); kinds["outliers"] = o
    want = {"uniform": 0, "gauss": 0, "powerlaw": _lib.I8_ROTATED, "outliers": _lib.I8_ROTATED}
    for name, x in kinds.items():
        tc = torch.from_numpy(x).cuda()
        fl, plain, rot = acc.choose_i8_layout(tc.data_ptr(), n, d)
        assert fl == want[name], (name, fl, plain, rot)
        t8 = torch.empty((_lib.i8_shadow_rows(n), d), dtype=torch.int8, device="cuda"); tm = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
        full_plain = acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm.data_ptr(), want_mean_err=True)
        full_rot = acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm.data_ptr(), want_mean_err=True, i8_flags=_lib.I8_ROTATED)
        assert abs(plain - full_plain) < 0.1 * full_plain and abs(rot - full_rot) < 0.1 * full_rot, (name, plain, full_plain, rot, full_rot)
        if name in ("powerlaw", "outliers"):
            assert full_rot < 0.3 * full_plain, (name, full_plain, full_rot)
    assert acc.choose_i8_layout(tc.data_ptr(), 100, d)[0] == _lib.I8_ROTATED           # (two blocks of the outlier rows still answer)


def test_anisotropic_corpus_stays_on_the_int8_tier_in_the_rotated_layout(acc, oracle):
    """The corpus of test_anisotropic_corpus_teaches_the_context_to_start_on_the_bf16_tier (a power-law spectrum: on the bench
    shard every query of an int8 batch escalates in the plain layout) with the shadow in the rotated layout: the bound is as
    tight as for isotropic rows, the batch is proven on the int8 tier (no escalation, no exhaustive fallback), the results
    are the oracle's and those of the plain layout."""
    rng = np.random.default_rng(72)
    n, d, nq, k = 200_000, 256, 160, 50
    scale = (np.arange(1, d + 1, dtype=np.float32) ** -0.5)
    x = (rng.standard_normal((n, d)).astype(np.float32) * scale); x /= np.linalg.norm(x, axis=1, keepdims=True)
    qs = (rng.standard_normal((nq, d)).astype(np.float32) * scale); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    r = check(acc, oracle, x, qs, k, max_queries=8, expect_path=0, shadow="i8", expect_tier=_lib.TIER_I8, i8_flags=_lib.I8_ROTATED)
    assert r.diag["escalated_queries"] == 0 and r.diag["exact_fallback_queries"] == 0, r.diag
    plain = run(acc, x, qs, k, shadow="i8", i8_flags=0)
    assert np.array_equal(plain.rows, r.rows) and np.array_equal(plain.scores.view(np.uint32), r.scores.view(np.uint32))
    # L2 over the same rows (raw queries through the same map), and a batch small enough for the resident-query form
    check(acc, oracle, x * np.float32(3.0), qs * np.float32(0.5), k, metric=SCAN_L2, max_queries=6, shadow="both", i8_flags=_lib.I8_ROTATED)
    check(acc, oracle, x, qs[:40], k, max_queries=6, shadow="i8", expect_tier=_lib.TIER_I8, i8_flags=_lib.I8_ROTATED)


def test_int8_tier_tests_pass_in_the_rotated_layout():
    """Every test of this file that exercises the int8 tier through run() / check(), once more with YAMS_TEST_I8_FLAGS=1: all
    their int8 shadows are built in the rotated layout (hostile rows, masks, ties, thresholds, L2 under every accumulate
    definition, clustered corpora, batches of every size) — the results must not notice."""
    import subprocess, sys
    if _I8_FLAGS:
        pytest.skip("this IS the rotated run")
    env = dict(os.environ, YAMS_TEST_I8_FLAGS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "int8 or i8 or hostile or clustered or tight_clusters or l2_under or golden or small_batches or resident"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def test_context_learns_how_deep_the_int8_proofs_of_a_corpus_go(oracle):
    """Rows with Gaussian components, 500k x 768: the int8 bound is wider than the gap between the 100th and the 385th best
    score, so no proof succeeds on the plan's 3k + 64 candidates — every query of the first batch is widened.  The context
    remembers that per corpus: the next batches re-score the whole list in stage 1 (nothing is widened, the same rows are
    re-scored once instead of twice); results are identical and the oracle's."""
    import torch
    from yams_amd.accel import Accel
    n, d, nq, k = 500_000, 768, 200, 100
    g = torch.Generator(device="cuda"); g.manual_seed(91)
    tc = torch.randn((n, d), generator=g, device="cuda"); tc /= tc.norm(dim=1, keepdim=True)
    tq = torch.randn((nq, d), generator=g, device="cuda"); tq /= tq.norm(dim=1, keepdim=True)
    torch.cuda.synchronize()
    qs = tq.cpu().numpy()
    a2 = Accel(0, torch.cuda.current_stream().cuda_stream)       # a context of its own: what it learns must not leak into other tests
    try:
        t8 = torch.empty((_lib.i8_shadow_rows(n), d), dtype=torch.int8, device="cuda"); tm = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
        a2.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm.data_ptr())
        v = a2.corpus_view(tc.data_ptr(), n, d, rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm.data_ptr())
        first = a2.scan_topk(v, qs, k, -1.0)
        assert first.diag["filter_tier"] == _lib.TIER_I8 and first.diag["widened_queries"] > nq // 2, first.diag
        later = a2.scan_topk(v, qs, k, -1.0)
        assert later.diag["filter_tier"] == _lib.TIER_I8 and later.diag["widened_queries"] == 0, later.diag
        assert later.diag["rescored_rows"] < first.diag["rescored_rows"] and later.diag["exact_fallback_queries"] == 0, (first.diag, later.diag)
        assert np.array_equal(first.rows, later.rows) and np.array_equal(first.scores.view(np.uint32), later.scores.view(np.uint32))
        x = tc.cpu().numpy()
        for qi in range(0, nq, 37):
            rows, sims, _, _ = oracle.scan_cosine(x, qs[qi], k, -1.0)
            assert np.array_equal(later.rows[qi], rows) and np.array_equal(later.scores[qi].view(np.uint32), sims.view(np.uint32)), qi
        small = a2.scan_topk(v, qs[:40], k, -1.0)                  # (smaller batches of the same corpus use what was learnt too)
        assert np.array_equal(small.rows, later.rows[:40]) and small.diag["widened_queries"] == 0, small.diag
    finally:
        a2.close()


def test_clustered_corpus_under_l2_takes_the_int8_second_pass(oracle):
    """Stage 2a under L2 (round 6): the filter's score is g = q.x - |x|^2 / 2, the second pass's threshold the g of the k-th
    smallest exact distance found — (|q|^2 - d_k^2) / 2, lowered by the proof's own margins (wider under fp32 accumulation).
    Queries near the centre of a cluster of ~1300 near-equidistant rows: the first pass's lists cannot prove a top-100, the
    second one does — no escalation to the bf16 tiers to speak of; rows, order, distances and similarities are the oracle's,
    under fp64 accumulation and under eight fp32 lanes.  Contexts of their own (what they learn about the corpus stays there)."""
    import torch
    from yams_amd.accel import Accel
    corpus, q = _clustered(300_000, 256, 230, 71, 200)
    corpus = (corpus * np.float32(1.7)).astype(np.float32); q = (q * np.float32(0.6)).astype(np.float32)
    a2 = Accel(0, torch.cuda.current_stream().cuda_stream)
    try:
        first = check(a2, oracle, corpus, q, 100, metric=SCAN_L2, max_queries=10, expect_path=0, shadow="both", expect_tier=_lib.TIER_I8)
        assert first.diag["retried_queries"] >= 10 and first.diag["escalated_queries"] <= first.diag["retried_queries"] // 3, first.diag
        assert first.diag["exact_fallback_queries"] == 0, first.diag
    finally:
        a2.close()
    a3 = Accel(0, torch.cuda.current_stream().cuda_stream)
    try:
        r = run(a3, corpus, q, 100, metric=SCAN_L2, flags=_lib.FLAG_L2_ACC_F32X8, shadow="both")
        assert r.diag["filter_tier"] == _lib.TIER_I8 and r.diag["retried_queries"] >= 10 and r.diag["exact_fallback_queries"] == 0, r.diag
        for qi in range(0, 200, 23):
            rows, dist, sims = oracle.scan_l2_f32acc(corpus, q[qi], 100, -1.0, lanes=8)
            assert np.array_equal(r.rows[qi], rows), qi
            assert np.array_equal(r.dist[qi].view(np.uint32), dist.view(np.uint32)) and np.array_equal(r.scores[qi].view(np.uint32), sims.view(np.uint32)), qi
    finally:
        a3.close()


def test_bf16_tier_takes_the_second_pass_and_learns_the_depth_too(oracle):
    """Dims that are not a multiple of 64 (here 304 = 19 x 16) stay on the single-pass bf16 tier; its lists are cut by the same sampled
    threshold, so clusters of ~1300 near-equal rows leave it as unproven as the int8 tier.  Round 6: the second pass serves
    this tier too (its score is within the error bound of the similarity either way: the threshold is the k-th exact score
    minus that bound), and the context learns the depth per tier: the next batch needs neither widening nor a second pass.
    Rows, order and score bits are the oracle's both times."""
    import torch
    from yams_amd.accel import Accel
    corpus, q = _clustered(300_000, 304, 230, 75, 200, spread=0.1)
    a2 = Accel(0, torch.cuda.current_stream().cuda_stream)
    try:
        first = check(a2, oracle, corpus, q, 100, max_queries=10, expect_path=0, shadow=True, expect_tier=2)
        assert first.diag["retried_queries"] >= 10 and first.diag["escalated_queries"] <= first.diag["retried_queries"] // 3, first.diag
        assert first.diag["exact_fallback_queries"] == 0, first.diag
        later = check(a2, oracle, corpus, q, 100, max_queries=4, expect_path=0, shadow=True, expect_tier=2)
        assert later.diag["widened_queries"] == 0 and later.diag["retried_queries"] <= first.diag["retried_queries"] // 4, (first.diag, later.diag)
        assert np.array_equal(first.rows, later.rows) and np.array_equal(first.scores.view(np.uint32), later.scores.view(np.uint32))
    finally:
        a2.close()
