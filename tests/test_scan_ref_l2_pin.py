"""The in-tree half of the L2 (vec0) path pinned by the REFERENCE'S OWN CODE, compiled here (round 6):
SqliteVecBackend::Impl::vec0SearchUnlocked (src/vector/sqlite_vec_backend.cpp:4450-4530), getVectorByRowidUnlocked
(:3084-3099), vec0TableName (:617-619) and the table's creation / population (ensureVec0TableUnlocked :3236-3249,
decodeVectorForDimRowUnlocked :3251-3267, rebuildVec0DimUnlocked :3350-3421), cut verbatim by oracle/gen_scan_ref.py and run
over an in-memory SQLite.  The `vec0` virtual table itself belongs to the ABSENT third_party/sqlite-vec-cpp: the harness
supplies a module whose distance function is pluggable (oracle/scan_ref_wrap.cpp) — so what is pinned here is everything
AROUND the distance: k nearest THEN the cosine threshold, rows the vectors table no longer holds, rows that never reach
the index (wrong blob size, non-finite), k > n, the candidate-rowid branch of the SQL, and the order equal distances
come back in (`ORDER BY distance` executed by SQLite over rowid-ordered input).  The distance ARITHMETIC stays
unpinned (calibrated at run time: include/yams_accel/l2_calibration.hpp): each test runs under the fp64 definition and
under fp32 definitions plugged into the module, against the oracle's function of the same name."""
import json
import os

import numpy as np
import pytest

import _cases
import _oracle


@pytest.fixture()
def table():
    t = _oracle.scan_ref()
    if t is None or not t.has_vec0:
        pytest.skip("oracle/_ref/libyams_scan_ref.so (with the vec0 doors) not present: built only where /root/reference exists")
    yield t
    t.close()


DEFS = [None, 1, 8, 16, -1, -8, -16]      # fp64; fp32 sequential / 8 / 16 lanes; the same with fused multiply-adds


def oracle_l2(oracle, lanes, corpus, q, k, thr):
    if lanes is None:
        rows, dist, sims = oracle.scan_l2(corpus, q, k, thr)
    else:
        rows, dist, sims = oracle.scan_l2_f32acc(corpus, q, k, thr, lanes=lanes)
    return rows, sims


def same(got, rows, sims):
    assert not isinstance(got, int), got
    ords, sc, _ = got
    assert np.array_equal(ords, rows), (ords[:12], rows[:12])
    assert np.array_equal(sc.view(np.uint32), sims.view(np.uint32)), (sc[:6], sims[:6])


@pytest.mark.parametrize("lanes", DEFS)
def test_vec0_search_k_then_threshold_under_every_distance_definition(oracle, table, lanes):
    """The reference's own recipe rows (mt19937, config 1's dimension), k = 1 / 10 / 100, thresholds off and on: the k
    nearest are taken FIRST, the cosine threshold then drops some of them (fewer than k come back, :4506-4516)."""
    n, d = 3000, 384
    corpus = oracle.mt19937_rows(42, 0, n, d)
    queries = oracle.mt19937_rows(42, n, 6, d)
    table.insert_rows(corpus)
    table.vec0_set_distance(lanes)
    table.vec0_rebuild(d)
    dropped = 0
    for qi in range(6):
        for k in (1, 10, 100):
            for thr in (-1.0, 0.1, 0.14):
                got = table.vec0_search(queries[qi], k, thr)
                rows, sims = oracle_l2(oracle, lanes, corpus, queries[qi], k, thr)
                same(got, rows, sims)
                assert got[2]["knn_queries"] == 1 and got[2]["rowid_probes"] == 0
                dropped += int(len(rows) < k)
    assert dropped > 0          # the threshold did bite somewhere: "k nearest, then the threshold" was exercised


def test_vec0_search_returns_equal_distances_in_rowid_order_whatever_the_chunk_ids(oracle, table):
    """Exact duplicates (equal distances under ANY arithmetic) with SHUFFLED chunk ids: the reference's statement is
    `ORDER BY distance` alone (:4473), SQLite's sorter keeps the order the rows came in — rowid order.  The oracle's L2
    functions break ties by row index (and ignore tie ranks); the cosine comparator's chunk_id rule (:4218-4223) does
    not apply to this path."""
    d = 32
    corpus = oracle.synth_rows(5, 0, 60, d)
    for r in (10, 17, 30, 44):
        corpus[r] = corpus[3]
    corpus[6] = corpus[5]
    corpus[50:55] = corpus[49]
    ids = ["c%04d" % ((37 * i) % 101) for i in range(60)]      # chunk-id order is NOT row order
    table.insert_rows(corpus, chunk_ids=ids)
    table.vec0_rebuild(d)
    rank = np.argsort(np.argsort(np.array(ids))).astype(np.uint64)
    for q in (corpus[3] + np.float32(0.01), corpus[49], oracle.synth_rows(5, 1000, 1, d)[0]):
        for k in (3, 5, 8, 60):     # k = 3: the cut falls INSIDE the group of five equal distances -> the smaller rowids stay
            got = table.vec0_search(q, k, -1.0)
            rows, dist, sims = oracle.scan_l2(corpus, q, k, -1.0)
            same(got, rows, sims)
            rows_r, _, _ = oracle.scan_l2(corpus, q, k, -1.0, tie_rank=rank)    # tie ranks change nothing under L2
            assert np.array_equal(rows_r, rows)
    got = table.vec0_search(corpus[3] + np.float32(0.01), 5, -1.0)
    assert list(got[0][:5]) == [3, 10, 17, 30, 44]


def test_vec0_search_skips_rows_the_vectors_table_no_longer_holds_and_rows_never_indexed(oracle, table):
    """(a) rows with a wrong-size blob or a non-finite component never reach the vec0 table (decodeVectorForDimRowUnlocked,
    :3251-3267); (b) a rowid the index still returns but `vectors` has lost (deleted after the rebuild) is skipped by
    getVectorByRowidUnlocked (:4501-4504) — and NOT replaced: the statement asked for k rows, fewer come back."""
    d = 24
    corpus = oracle.synth_rows(9, 0, 200, d)
    bad_nan, bad_inf = 20, 77
    corpus[bad_nan, 3] = np.nan
    corpus[bad_inf, 0] = np.inf
    for r in range(200):
        if r == 120:        # a row of the WRONG size (half a vector) that claims the right dimension
            table.insert_raw("c%018d" % r, corpus[r, :d // 2].tobytes(), d)
        else:
            table.insert_raw("c%018d" % r, corpus[r].tobytes(), d)
    table.vec0_rebuild(d)
    usable = np.ones(200, bool); usable[[bad_nan, bad_inf, 120]] = False
    sub = corpus[usable]; back = np.flatnonzero(usable)
    q = oracle.synth_rows(9, 5000, 3, d)
    for qi in range(3):
        got = table.vec0_search(q[qi], 200, -1.0)
        rows, dist, sims = oracle.scan_l2(sub, q[qi], 200, -1.0)
        same(got, back[rows], sims)
        assert len(rows) == 197
    # (b): delete three of query 0's ten nearest from `vectors` only
    rows10, _, sims10 = oracle.scan_l2(sub, q[0], 10, -1.0)
    victims = [int(back[rows10[i]]) for i in (0, 4, 9)]
    for v in victims:
        table.delete_ordinal(v)
    got = table.vec0_search(q[0], 10, -1.0)
    keep = [i for i in range(10) if int(back[rows10[i]]) not in victims]
    same(got, back[rows10[keep]], sims10[keep])
    assert len(got[0]) == 7


@pytest.mark.parametrize("lanes", [None, -8])
def test_vec0_search_with_k_above_n_k_zero_and_empty_queries(oracle, table, lanes):
    d = 16
    corpus = oracle.synth_rows(3, 0, 7, d)
    table.insert_rows(corpus)
    table.vec0_set_distance(lanes)
    table.vec0_rebuild(d)
    q = oracle.synth_rows(3, 100, 1, d)[0]
    same(table.vec0_search(q, 100, -1.0), *oracle_l2(oracle, lanes, corpus, q, 100, -1.0))
    assert len(table.vec0_search(q, 100, -1.0)[0]) == 7
    got = table.vec0_search(q, 0, -1.0)                 # k == 0 -> empty before anything is prepared (:4453-4455)
    assert len(got[0]) == 0 and got[2]["knn_queries"] == 0
    assert len(oracle.scan_l2(corpus, q, 0, -1.0)[0]) == 0
    got = table.vec0_search(q, 5, -1.0, candidate_rowids=[])   # an empty candidate list -> empty (:4456-4458)
    assert len(got[0]) == 0 and got[2]["knn_queries"] == 0 and got[2]["rowid_probes"] == 0


def test_vec0_search_candidate_rowid_branch(oracle, table):
    """`AND rowid IN (SELECT value FROM json_each(?3))` (:4469-4472, :4491-4495): the candidate list travels as a JSON array
    and the search is restricted to those rowids.  This image's SQLite (3.36, no sqlite3_vtab_in) hands the module one
    rowid per probe, so the k cut is the reference loop's own (`records.size() >= k`, :4514-4517): identical to a KNN
    among the candidates whenever no hit is dropped by the threshold — the cases below — or k >= the candidate count."""
    n, d = 500, 48
    corpus = oracle.synth_rows(21, 0, n, d)
    corpus[40] = corpus[7]; corpus[300] = corpus[7]        # equal distances among the candidates too
    table.insert_rows(corpus)
    table.vec0_rebuild(d)
    q = oracle.synth_rows(21, 9000, 4, d)
    rng = np.random.default_rng(4)
    for qi in range(4):
        cand = np.sort(rng.choice(n, 90, replace=False))
        if qi == 0:
            cand = np.union1d(cand, [7, 40, 300])
        rowids = [table.rowid_of(int(c)) for c in cand]
        assert rowids == [int(c) + 1 for c in cand]         # (the mirror's order IS rowid order)
        for k in (5, 90, 200):
            got = table.vec0_search(q[qi] if qi else corpus[7] + np.float32(0.02), k, -1.0, candidate_rowids=rowids + [10_000_000])  # an unknown rowid is no row
            rows, dist, sims = oracle.scan_l2(corpus[cand], q[qi] if qi else corpus[7] + np.float32(0.02), k, -1.0)
            same(got, cand[rows], sims)
            assert got[2]["knn_queries"] == 0 and got[2]["rowid_probes"] == len(rowids) + 1
        # with a threshold, k >= the candidate count: every candidate is looked at, the threshold filters, order stays
        got = table.vec0_search(q[qi], 200, 0.0, candidate_rowids=rowids)
        rows, dist, sims = oracle.scan_l2(corpus[cand], q[qi], 200, 0.0)
        same(got, cand[rows], sims)


GOLDEN_LANES = {"f64": None, "f32": 1, "f32x8": 8, "f32x16": 16, "f32_fma": -1, "f32x8_fma": -8, "f32x16_fma": -16}


def golden_l2_expected(case, name):
    return case["expected"]["f64" if name in case["same_as_f64"] else name]


def test_oracle_reproduces_the_reference_compiled_l2_golden_vectors(oracle):
    """tests/golden/scan_l2.json (written by tests/golden/make_scan_l2_golden.py from the reference's compiled
    vec0SearchUnlocked, one result per plugged distance definition) against oracle/yams_oracle.c — runs wherever the
    committed file is, with or without /root/reference."""
    with open(os.path.join(_cases.GOLDEN, "scan_l2.json")) as f:
        g = json.load(f)
    n = 0
    for case in g["cases"]:
        corpus, queries, tie_rank, allow = _cases.golden_scan_inputs(oracle, case)
        sub, back = corpus, np.arange(corpus.shape[0])
        if allow is not None:
            back = np.flatnonzero(allow); sub = corpus[back]
        for name, lanes in GOLDEN_LANES.items():
            exp = golden_l2_expected(case, name)
            for qi, e in enumerate(exp):
                rows, sims = oracle_l2(oracle, lanes, sub, queries[qi], case["k"], case["threshold"])
                assert back[rows].tolist() == e["rows"], (case["name"], name, qi)
                assert [int(x) for x in sims.view(np.uint32)] == e["score_bits"], (case["name"], name, qi)
                n += 1
    assert n >= 300
