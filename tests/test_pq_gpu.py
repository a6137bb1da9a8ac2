"""The product-quantised engine on the device: yams_scan_pq_topk_device — ADC scan over host-supplied codes
and per-query tables, best approxK by (score desc, tie key asc), exact re-rank, final (similarity desc, chunk_id asc) —
against oracle_pq_search, the restatement of SqliteVecBackend::Impl::simeonPqSearchUnlocked (sqlite_vec_backend.cpp:3868-4056).
PARITY UNPINNED for the order of the ADC sum (third_party/simeon is absent): all four served orders are checked against the
oracle's function of the same name.  Rows, order and score bits must be identical."""
import numpy as np
import pytest

import _pq

pytestmark = pytest.mark.gpu


def build(oracle, n, d, m, seed, scale=True, dups=()):
    corpus = oracle.synth_rows(seed, 0, n, d)
    if scale:
        corpus = (corpus * np.linspace(0.25, 4.0, n, dtype=np.float32)[:, None]).astype(np.float32)    # raw rows, not unit (:3002-3003)
    for lo, hi, src in dups:
        corpus[lo:hi] = corpus[src]
    u = _pq.unit(corpus)
    pq = _pq.Pq(u, m, seed)
    codes = pq.encode(u)
    ids = ["c%07d" % ((7919 * i) % 1000003) for i in range(n)]
    keys = np.array([_pq.stable_string_key(s) for s in ids], np.uint64)
    rank = np.argsort(np.argsort(np.array(ids))).astype(np.uint32)
    return corpus, pq, codes, keys, rank


def check(acc, oracle, corpus, pq, codes, keys, rank, queries, k, thr=-1.0, rf=2, lanes=1, roi=None, cand=None):
    luts = np.stack([pq.lut(q) for q in queries])
    d_rows = acc.to_device(corpus)
    bufs = [d_rows]
    tie = inv = None
    if rank is not None:
        invp = np.empty_like(rank); invp[rank] = np.arange(rank.size, dtype=rank.dtype)
        tie, inv = acc.to_device(rank.astype(np.uint32)), acc.to_device(invp.astype(np.uint32)); bufs += [tie, inv]
    v = acc.corpus_view(d_rows.ptr, corpus.shape[0], corpus.shape[1], tie_rank_ptr=tie.ptr if tie else None, rank_row_ptr=inv.ptr if inv else None)
    try:
        r = acc.scan_pq_topk(v, codes, luts, queries, k, thr, rf, tie_keys=keys, row_of_index=roi, candidates=cand, sum_lanes=lanes)
    finally:
        for b in bufs:
            b.free()
    for qi, q in enumerate(queries):
        rows, sims, st = oracle.pq_search(corpus, codes, luts[qi], q, k, thr, rf, tie_keys=keys, row_of_index=roi,
                                          chunk_rank=rank.astype(np.uint64) if rank is not None else None, candidates=cand, sum_lanes=lanes)
        cnt = int(r.counts[qi])
        assert cnt == len(rows), (qi, cnt, len(rows), r.diag)
        assert r.rows[qi, :cnt].tolist() == rows.tolist(), (qi, r.diag)
        assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), qi
    return r


@pytest.mark.parametrize("lanes", [1, 4, 8, 16])
def test_pq_search_matches_the_oracle_under_every_sum_order(acc, oracle, lanes):
    """The reference's default shape — dim 384, 32 sub-quantisers, rerank factor 2, top-10 / top-100 — on raw (un-normalised)
    rows with runs of duplicates (equal ADC scores: the tie key decides; equal exact similarities: the chunk id decides)."""
    corpus, pq, codes, keys, rank = build(oracle, 30_000, 384, 32, 11, dups=[(100, 140, 100), (9000, 9008, 100)])
    queries = np.concatenate([oracle.synth_rows(11, 1 << 40, 5, 384) * np.float32(2.5), corpus[100:101] * np.float32(0.7)])
    r = check(acc, oracle, corpus, pq, codes, keys, rank, queries, 10, lanes=lanes)
    assert r.diag["path"] == 2 and r.diag["filter_tier"] == 5 and r.diag["rows_visited"] == 6 * 30_000
    assert r.diag["exact_distance_evaluations"] == 6 * 20
    check(acc, oracle, corpus, pq, codes, keys, rank, queries, 100, lanes=lanes)


def test_pq_search_threshold_rerank_factors_and_k_above_the_index(acc, oracle):
    corpus, pq, codes, keys, rank = build(oracle, 5_000, 256, 16, 12)
    queries = oracle.synth_rows(12, 1 << 40, 4, 256)
    for k, rf, thr in ((10, 1, -1.0), (10, 8, -1.0), (50, 4, 0.15), (7, 2, 0.5), (1000, 2, -1.0)):
        check(acc, oracle, corpus, pq, codes, keys, rank, queries, k, thr, rf)
    small = build(oracle, 37, 64, 8, 13)
    check(acc, oracle, *small, oracle.synth_rows(13, 1 << 40, 3, 64), 100)          # k above the index: everything, re-ranked


def test_pq_search_candidate_indices_and_rows_the_table_lost(acc, oracle):
    """The candidate restriction (:3910-3937: ascending indices of the named documents' rows) and indexed rows whose
    vectors-table row is gone (:4010-4012: skipped, NOT replaced); index order != row order (row_of_index)."""
    corpus, pq, codes, keys, rank = build(oracle, 8_000, 128, 16, 14, dups=[(10, 30, 10)])
    n = corpus.shape[0]
    queries = oracle.synth_rows(14, 1 << 40, 3, 128)
    perm = np.random.default_rng(14).permutation(n).astype(np.uint32)       # index i of the PQ arrays is corpus row perm[i]
    codes_p, keys_p = codes[perm], keys[perm]
    roi = perm.copy()
    lost = np.flatnonzero(np.isin(perm, [10, 11, 12, 500]))                  # their rows are gone from the vectors table
    roi[lost] = n + 3
    cand = np.sort(np.random.default_rng(15).choice(n, 900, replace=False)).astype(np.uint32)
    check(acc, oracle, corpus, pq, codes_p, keys_p, rank, queries, 20, roi=roi, cand=cand)
    check(acc, oracle, corpus, pq, codes_p, keys_p, rank, queries, 20, roi=roi)
    # an empty candidate list returns nothing
    luts = np.stack([pq.lut(q) for q in queries])
    d_rows = acc.to_device(corpus)
    v = acc.corpus_view(d_rows.ptr, n, 128)
    r = acc.scan_pq_topk(v, codes_p, luts, queries, 5, candidates=np.zeros(0, np.uint32), row_of_index=roi)
    assert r.counts.tolist() == [0, 0, 0]
    d_rows.free()


def test_pq_search_refuses_what_the_hosts_normalisation_refuses(acc, oracle):
    """norm^2 <= 1e-20 (normalizeEmbeddingInPlace, :213-226) -> an empty result for that query, the others are served."""
    corpus, pq, codes, keys, rank = build(oracle, 4_000, 64, 8, 16)
    good = oracle.synth_rows(16, 1 << 40, 2, 64)
    queries = np.stack([good[0], np.zeros(64, np.float32), np.full(64, 1e-12, np.float32), good[1] * np.float32(1e-8)])
    luts = np.stack([pq.lut(q) if float((q.astype(np.float64) ** 2).sum()) > 1e-20 else np.zeros((8, 256), np.float32) for q in queries])
    d_rows = acc.to_device(corpus)
    v = acc.corpus_view(d_rows.ptr, 4_000, 64)
    r = acc.scan_pq_topk(v, codes, luts, queries, 5, tie_keys=keys)
    d_rows.free()
    assert r.counts.tolist() == [5, 0, 0, 5]
    for qi in (0, 3):
        rows, sims, _ = oracle.pq_search(corpus, codes, luts[qi], queries[qi], 5, tie_keys=keys)
        assert r.rows[qi, :5].tolist() == rows.tolist() and np.array_equal(r.scores[qi, :5].view(np.uint32), sims.view(np.uint32))


def test_pq_search_at_the_reference_default_scale(acc, oracle):
    """1M codes x 32 sub-quantisers, 64 queries (one ADC batch, several re-rank batches): every query against the oracle."""
    n, d, m = 1_000_000, 384, 32
    rng = np.random.default_rng(17)
    corpus = oracle.synth_rows(17, 0, 20_000, d)                              # the rows the re-rank reads: 20k distinct rows ...
    roi = rng.integers(0, 20_000, n).astype(np.uint32)                       # ... that 1M index entries point at
    pq = _pq.Pq(_pq.unit(corpus), m, 17)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)                    # (random codes: the scan does not care where they came from)
    keys = rng.integers(0, 1 << 63, n).astype(np.uint64)
    queries = oracle.synth_rows(17, 1 << 40, 64, d)
    luts = np.stack([pq.lut(q) for q in queries])
    d_rows = acc.to_device(corpus)
    v = acc.corpus_view(d_rows.ptr, corpus.shape[0], d)
    r = acc.scan_pq_topk(v, codes, luts, queries, 10, -1.0, 2, tie_keys=keys, row_of_index=roi)
    d_rows.free()
    for qi in range(0, 64, 7):
        rows, sims, _ = oracle.pq_search(corpus, codes, luts[qi], queries[qi], 10, -1.0, 2, tie_keys=keys, row_of_index=roi)
        cnt = int(r.counts[qi])
        assert cnt == len(rows) and r.rows[qi, :cnt].tolist() == rows.tolist(), qi
        assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), qi


def _random_index(rng, n, m, d, n_rows=8_000):
    """Random codes over n index entries that point at n_rows distinct rows (the scan does not care where codes came from)."""
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    keys = rng.integers(0, 1 << 63, n).astype(np.uint64)
    roi = rng.integers(0, n_rows, n).astype(np.uint32)
    return codes, keys, roi


@pytest.mark.parametrize("m,lanes,k,rf", [(32, 1, 100, 2), (32, 8, 10, 4), (8, 4, 50, 2), (40, 16, 100, 2), (100, 1, 20, 2), (7, 1, 30, 3), (32, 1, 1000, 2)])
def test_filtered_adc_scan_lists_exactly_the_best(acc, oracle, m, lanes, k, rf):
    """Indexes of >= 65 536 entries go through the FILTERED form (pq_adc_filter_kernel: sample -> threshold -> keys of the codes
    that reach it): the best approxK of each list must be the best approxK of all codes — rows, order and score bits equal the
    oracle's — under every served sum order, every table-group size (m <= 36: four queries' tables per workgroup, <= 72: two,
    else one), an m that is not a multiple of four, approxK up to 2000."""
    n, d = 150_001, (m * 8 if m % 4 == 0 else m * 16)
    rng = np.random.default_rng(300 + m + lanes)
    corpus = oracle.synth_rows(300 + m, 0, 8_000, d)
    pq = _pq.Pq(_pq.unit(corpus), m, 300 + m)
    codes, keys, roi = _random_index(rng, n, m, d)
    codes[500:560] = codes[499]                                              # a run of equal ADC scores: the tie key decides
    queries = oracle.synth_rows(300 + m, 1 << 40, 9, d)
    luts = np.stack([pq.lut(q) for q in queries])
    d_rows = acc.to_device(corpus)
    v = acc.corpus_view(d_rows.ptr, corpus.shape[0], d)
    r = acc.scan_pq_topk(v, codes, luts, queries, k, -1.0, rf, tie_keys=keys, row_of_index=roi, sum_lanes=lanes)
    d_rows.free()
    assert r.diag["exact_fallback_queries"] == 0, r.diag                     # every query was served by the filtered form
    for qi in range(9):
        rows, sims, _ = oracle.pq_search(corpus, codes, luts[qi], queries[qi], k, -1.0, rf, tie_keys=keys, row_of_index=roi, sum_lanes=lanes)
        cnt = int(r.counts[qi])
        assert cnt == len(rows) and r.rows[qi, :cnt].tolist() == rows.tolist(), (qi, r.diag)
        assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), qi


def test_filtered_adc_scan_falls_back_where_its_lists_are_no_use(acc, oracle):
    """Tables that make every code score the same (all zeros: the threshold is that score, every code reaches it, the list is
    cut off), a table with infinite entries (checked on the codes whose score is a number: a NaN score sorts nowhere in the
    reference's comparator — undefined there, left out here) and a table whose best scores tie by the thousand:
    those queries are served by the unfiltered form, the others by the filtered one, all identical to the oracle.  Plus a
    candidate restriction of 70 000 indices (filtered) and one of 3 000 (unfiltered)."""
    n, m, d, k = 120_000, 16, 128, 25
    rng = np.random.default_rng(321)
    corpus = oracle.synth_rows(321, 0, 8_000, d)
    pq = _pq.Pq(_pq.unit(corpus), m, 321)
    codes, keys, roi = _random_index(rng, n, m, d)
    queries = oracle.synth_rows(321, 1 << 40, 6, d)
    luts = np.stack([pq.lut(q) for q in queries])
    luts[1] = 0.0                                                            # every score 0.0
    luts[3, 2, 17] = np.inf; luts[3, 9, 200] = -np.inf                       # infinite entries: scores of +inf and -inf ...
    codes[(codes[:, 2] == 17) & (codes[:, 9] == 200), 9] = 201               # ... but no code with both (NaN: undefined in the reference)
    luts[4] = np.round(luts[4] * 4) / 4                                      # a coarse table: thousands of equal scores
    d_rows = acc.to_device(corpus)
    v = acc.corpus_view(d_rows.ptr, corpus.shape[0], d)
    for cand in (None, np.sort(rng.choice(n, 70_000, replace=False)).astype(np.uint32), np.sort(rng.choice(n, 3_000, replace=False)).astype(np.uint32)):
        r = acc.scan_pq_topk(v, codes, luts, queries, k, -1.0, 2, tie_keys=keys, row_of_index=roi, candidates=cand)
        if cand is None or cand.size >= 65536:
            assert 1 <= r.diag["exact_fallback_queries"] <= 3, r.diag        # the all-equal table for certain
        else:
            assert r.diag["exact_fallback_queries"] == 6, r.diag
        for qi in range(6):
            rows, sims, _ = oracle.pq_search(corpus, codes, luts[qi], queries[qi], k, -1.0, 2, tie_keys=keys, row_of_index=roi, candidates=cand)
            cnt = int(r.counts[qi])
            assert cnt == len(rows) and r.rows[qi, :cnt].tolist() == rows.tolist(), (qi, cand is None, r.diag)
            assert np.array_equal(r.scores[qi, :cnt].view(np.uint32), sims.view(np.uint32)), qi
    d_rows.free()
