"""The cosine scan's oracle pinned by the REFERENCE'S OWN CODE, compiled here: oracle/_ref/libyams_scan_ref.so holds
SqliteVecBackend::Impl::bruteForceSearchUnlocked (src/vector/sqlite_vec_backend.cpp:4115-4409, fast path AND the
metadata-filter record path), its helpers (:204-236), recordFromStatement (:3102-3160) and
VectorDatabase::computeCosineSimilarity (src/vector/vector_database.cpp:1786-1810), cut verbatim out of /root/reference by
oracle/gen_scan_ref.py and run over an in-memory SQLite `vectors` table (:342-371).  Every test feeds the same rows and
queries to that code and to oracle/yams_oracle.c (oracle_exact_scan_cosine / _records / oracle_cosine_similarity) and
demands identical rows, order, score BITS and diagnostics.  Where the library did not travel (no /root/reference and no
prebuilt file) the live tests skip and the committed golden file (tests/golden/scan.json, written by
tests/golden/make_scan_golden.py from the same library) pins the oracle instead."""
import json
import os

import numpy as np
import pytest

import _cases
import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "scan.json")


@pytest.fixture()
def table():
    t = _oracle.scan_ref()
    if t is None:
        pytest.skip("oracle/_ref/libyams_scan_ref.so not present (built only where /root/reference exists)")
    yield t
    t.close()


def same(ref_out, rows, sims):
    ords, sc, _ = ref_out
    assert np.array_equal(ords, rows), (ords[:10], rows[:10])
    assert np.array_equal(sc.view(np.uint32), sims.view(np.uint32)), (sc[:6], sims[:6])


def test_reference_loop_on_the_mt19937_config1_recipe(oracle, table):
    """BASELINE config 1: 10k x 384, the reference's own embedding recipe (vector_backend_engine_compare.cpp:83-107),
    top-10, 16 queries, thresholds on and off: rows, order, score bits, rowsVisited, exactDistanceEvaluations."""
    n, d, k = 10_000, 384, 10
    corpus = oracle.mt19937_rows(42, 0, n, d)
    queries = oracle.mt19937_rows(42, n, 16, d)
    table.insert_rows(corpus)
    for qi in range(16):
        for thr in (-1.0, 0.0, 0.05, 0.12):
            r = table.search(queries[qi], k, thr)
            rows, sims, visited, evals = oracle.scan_cosine(corpus, queries[qi], k, thr)
            same(r, rows, sims)
            assert r[2]["rows_visited"] == visited == n and r[2]["exact_distance_evaluations"] == evals == n
            assert r[2]["returned_rows"] == len(rows) and r[2]["used_exact_scan"] == 1


def test_reference_loop_with_k_above_n_k_zero_and_invalid_queries(oracle, table):
    d = 16
    corpus = oracle.synth_rows(3, 0, 7, d)
    table.insert_rows(corpus)
    q = oracle.synth_rows(3, 100, 1, d)[0]
    r = table.search(q, 100, -1.0)
    rows, sims, _, _ = oracle.scan_cosine(corpus, q, 100, -1.0)
    same(r, rows, sims)
    assert len(rows) == 7
    # k == 0 -> empty BEFORE the query is validated (:4123-4126)
    ords, sc, dg = table.search(np.zeros(d, np.float32), 0, -1.0)
    assert len(ords) == 0 and dg["used_exact_scan"] == 0
    assert list(oracle.scan_cosine(corpus, np.zeros(d, np.float32), 0, -1.0)[0]) == []
    # zero-norm (norm^2 < 1e-10), NaN and inf queries are InvalidArgument (:4127-4130); 1.1e-5 * e0 is just valid
    for bad in (np.zeros(d, np.float32), np.full(d, np.nan, np.float32), np.r_[np.inf, np.zeros(d - 1)].astype(np.float32),
                np.r_[9e-6, np.zeros(d - 1)].astype(np.float32)):
        assert table.search(bad, 3, -1.0) == table.invalid_argument
        assert oracle.scan_cosine(corpus, bad, 3, -1.0) is None
    ok = np.r_[1.1e-5, np.zeros(d - 1)].astype(np.float32)
    r = table.search(ok, 3, -1.0)
    assert not isinstance(r, int)
    same(r, *oracle.scan_cosine(corpus, ok, 3, -1.0)[:2])


def test_reference_loop_skips_and_extremes(oracle, table):
    """Rows the loop skips (:4244-4246 wrong blob size, :4258-4269 non-finite / norm^2 <= 1e-12) and the +-FLT_MAX/4 rows of
    vector_smoke_catch2_test.cpp:263-302, in one table; the oracle sees the same rows minus the ones that cannot be
    expressed as a dense matrix (wrong-size blobs: replaced by NaN rows, which are skipped as well but counted as
    evaluated — so evaluations are compared after the adjustment the reference makes, :4248-4251)."""
    d = 8
    L = np.float32(np.finfo(np.float32).max / 4)
    rows = [np.array([1, 0, 0, 0, 0, 0, 0, 0], np.float32), np.zeros(d, np.float32),
            np.array([np.nan, 1, 0, 0, 0, 0, 0, 0], np.float32), np.array([1e-7, 0, 0, 0, 0, 0, 0, 0], np.float32),
            np.array([2, 0, 0, 0, 0, 0, 0, 0], np.float32), np.array([L, -L, L, -L, L, -L, L, -L], np.float32),
            np.array([-L, L, -L, L, -L, L, -L, L], np.float32), np.array([0, np.inf, 0, 0, 0, 0, 0, 0], np.float32),
            np.array([1.1e-6, 0, 0, 0, 0, 0, 0, 0], np.float32),         # norm^2 = 1.21e-12: just above the fast path's bound
            np.array([0.5, 0.5, 0, 0, 0, 0, 0, 0], np.float32)]
    dense = np.stack(rows)
    for i, r in enumerate(rows):
        table.insert_raw("c%04d" % i, r.tobytes(), d)
    table.insert_raw("c%04d" % len(rows), np.ones(d - 1, np.float32).tobytes(), d)     # wrong size: skipped before it is evaluated
    table.insert_raw("c%04d" % (len(rows) + 1), None, d)                              # NULL blob: likewise
    for q in (rows[0], rows[5], rows[9], np.array([1, 1, 1, 1, 1, 1, 1, 1], np.float32), np.array([L, L, L, L, L, L, L, L], np.float32)):
        for thr in (-1.0, 0.0, 0.5):
            r = table.search(q, 20, thr)
            o_rows, o_sims, visited, evals = oracle.scan_cosine(dense, q, 20, thr)
            same(r, o_rows, o_sims)
            assert r[2]["rows_visited"] == len(rows) + 2 and r[2]["exact_distance_evaluations"] == evals == len(rows)


def test_reference_loop_breaks_ties_by_chunk_id(oracle, table):
    """Equal scores are ordered by chunk_id (:4218-4223, :4289-4306), whatever the insertion order: 40 rows in 5 groups of
    identical vectors under shuffled string ids, every k from 1 to 40 (the heap's replacement rule at every fill level)."""
    rng = np.random.default_rng(11)
    d = 12
    protos = rng.standard_normal((5, d)).astype(np.float32)
    which = rng.integers(0, 5, 40)
    corpus = protos[which]
    ids = ["id_%03d" % v for v in rng.permutation(40)]
    rank, _ = _cases.string_ranks(ids)
    table.insert_rows(corpus, chunk_ids=ids)
    q = (protos[0] + 0.3 * protos[1]).astype(np.float32)
    for k in range(1, 41):
        r = table.search(q, k, -1.0)
        rows, sims, _, _ = oracle.scan_cosine(corpus, q, k, -1.0, tie_rank=rank)
        same(r, rows, sims)


def test_reference_record_path_and_cosine_helper(oracle, table):
    """The metadata-filter path (:4333-4409): only rows whose metadata matches, computeCosineSimilarity scores, the
    1e-10 zero-norm rule, full sort, TopK and AllMatching; and the helper itself on adversarial pairs."""
    rng = np.random.default_rng(21)
    n, d = 600, 24
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[5] = 0.0
    corpus[6] = np.r_[2e-6, np.zeros(d - 1)]            # norm^2 = 4e-12: dropped by this path (< 1e-10), kept by the fast path
    corpus[7, 3] = np.nan
    corpus[100:110] = corpus[100]                        # a run of ties
    ids = ["m%04d" % v for v in rng.permutation(n)]
    rank, _ = _cases.string_ranks(ids)
    allow = np.zeros(n, np.uint8)
    for i in range(n):
        tag = "a" if i % 3 == 0 else "b"
        allow[i] = tag == "a"
        table.insert_raw(ids[i], corpus[i].tobytes(), d, metadata={"tag": tag, "n": str(i % 2)})
    q = rng.standard_normal(d).astype(np.float32)
    for k, thr in ((1, -1.0), (10, -1.0), (50, 0.1), (1000, -1.0)):
        r = table.search(q, k, thr, metadata_filters={"tag": "a"})
        rows, sims, ev = oracle.scan_cosine_records(corpus, q, k, thr, tie_rank=rank, allow=allow)
        same(r, rows, sims)
        assert r[2]["exact_distance_evaluations"] == ev and r[2]["rows_visited"] == n
    r = table.search(q, 0, 0.0, metadata_filters={"tag": "a"}, all_matching=True)
    rows, sims, ev = oracle.scan_cosine_records(corpus, q, 0, 0.0, tie_rank=rank, allow=allow, all_matching=True)
    same(r, rows, sims)
    both = allow & (np.arange(n) % 2 == 1)
    r = table.search(q, 7, -1.0, metadata_filters={"tag": "a", "n": "1"})
    same(r, *oracle.scan_cosine_records(corpus, q, 7, -1.0, tie_rank=rank, allow=both.astype(np.uint8))[:2])
    # AllMatching on the fast path (:4281-4287, :4315-4317)
    t2 = _oracle.scan_ref()
    t2.insert_rows(corpus, chunk_ids=ids)
    r = t2.search(q, 0, 0.2, all_matching=True)
    o_rows, o_sims, _, _ = oracle.scan_cosine(corpus, q, n, 0.2, tie_rank=rank)
    same(r, o_rows, o_sims)
    t2.close()
    # computeCosineSimilarity: sizes differ -> 0; zero vector -> 0; large magnitudes; denormals
    L = np.float32(np.finfo(np.float32).max / 4)
    pairs = [(corpus[1], corpus[2]), (corpus[5], corpus[2]), (np.full(d, L, np.float32), np.full(d, -L, np.float32)),
             (np.full(d, 1e-40, np.float32), np.full(d, 1e-40, np.float32)), (corpus[100], corpus[101])]
    for a, b in pairs:
        assert table.cosine(a, b) == oracle.cosine(a, b)
    assert table.cosine(corpus[1], corpus[2][:5]) == 0.0


def test_golden_scan_vectors_pin_the_oracle(oracle):
    """tests/golden/scan.json (generated by tests/golden/make_scan_golden.py from the reference-compiled loop): the oracle
    reproduces every recorded case — this is the pin that travels to the GPU box."""
    with open(GOLDEN) as f:
        g = json.load(f)
    assert g["generator"].endswith("make_scan_golden.py") and g["reference_spans"]
    for case in g["cases"]:
        corpus, queries, tie_rank, allow = _cases.golden_scan_inputs(oracle, case)
        for qi, exp in enumerate(case["expected"]):
            if case["path"] == "fast" and allow is not None:      # a candidate set: the oracle over the allowed rows, mapped back
                idx = np.flatnonzero(allow)
                sub_rank = None if tie_rank is None else np.argsort(np.argsort(tie_rank[idx])).astype(np.uint32)
                out = oracle.scan_cosine(corpus[idx], queries[qi], case["k"], case["threshold"], tie_rank=sub_rank)
                out = (idx[out[0]], out[1])
            elif case["path"] == "fast":
                out = oracle.scan_cosine(corpus, queries[qi], case["k"], case["threshold"], tie_rank=tie_rank)
            else:
                out = oracle.scan_cosine_records(corpus, queries[qi], case["k"], case["threshold"], tie_rank=tie_rank, allow=allow,
                                                 all_matching=case.get("all_matching", False))
            if exp.get("error"):
                assert out is None
                continue
            assert list(out[0]) == exp["rows"], (case["name"], qi)
            assert [int(x) for x in out[1].view(np.uint32)] == exp["score_bits"], (case["name"], qi)


def test_reference_candidate_and_document_restriction(oracle, table):
    """`document_hash` / `candidate_hashes` restrict the rows the reference's statement VISITS (pushed into SQL,
    :4151-4195): first the reference's own known answer (vector_smoke_catch2_test.cpp:355-401 — 2 rows visited, order
    allowed_best, allowed_second), then 900 rows over 30 documents: the restricted search equals the oracle over exactly
    the rows of the named documents, in their order, and rowsVisited / exactDistanceEvaluations count those rows only —
    the contract the host mirror's allow-mask implements (include/yams_accel/vector_index.hpp)."""
    t = table
    t.insert_raw("allowed_best", np.array([1, 0, 0, 0], np.float32).tobytes(), 4, document_hash="allowed")
    t.insert_raw("allowed_second", np.array([0.8, 0.6, 0, 0], np.float32).tobytes(), 4, document_hash="allowed")
    t.insert_raw("blocked", np.array([1, 0, 0, 0], np.float32).tobytes(), 4, document_hash="blocked")
    ords, sc, dg = t.search(np.array([1, 0, 0, 0], np.float32), 4, -1.0, candidate_hashes={"allowed"})
    assert list(ords) == [0, 1] and dg["rows_visited"] == 2 and dg["exact_distance_evaluations"] == 2 and dg["returned_rows"] == 2
    rng = np.random.default_rng(33)
    n, d, k = 900, 32, 15
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[40:44] = corpus[40]                                        # ties inside and across documents
    corpus[700] = corpus[40]
    docs = ["doc_%02d" % (i % 30) for i in range(n)]
    ids = ["x%05d" % v for v in rng.permutation(n)]
    rank, _ = _cases.string_ranks(ids)
    t2 = _oracle.scan_ref()
    t2.insert_rows(corpus, chunk_ids=ids, document_hashes=docs)
    q = corpus[40] + 0.1 * rng.standard_normal(d).astype(np.float32)
    for doc, cands in ((None, {"doc_10", "doc_03", "doc_21"}), ("doc_10", None), ("doc_10", {"doc_10", "doc_11"}), ("doc_10", {"doc_11"}),
                       (None, {"no_such_doc"})):
        allowed = np.array([(doc is None or docs[i] == doc) and (cands is None or docs[i] in cands) for i in range(n)])
        idx = np.flatnonzero(allowed)
        for thr in (-1.0, 0.1):
            r = t2.search(q, k, thr, document_hash=doc, candidate_hashes=cands)
            if len(idx) == 0:
                assert len(r[0]) == 0 and r[2]["rows_visited"] == 0
                continue
            sub_rank, _ = _cases.string_ranks([ids[i] for i in idx])
            rows, sims, visited, evals = oracle.scan_cosine(corpus[idx], q, k, thr, tie_rank=sub_rank)
            assert np.array_equal(r[0], idx[rows]), (doc, cands, r[0][:6], idx[rows][:6])
            assert np.array_equal(r[1].view(np.uint32), sims.view(np.uint32))
            assert r[2]["rows_visited"] == visited == len(idx) and r[2]["exact_distance_evaluations"] == evals
    # ... and through the metadata path with a candidate set: the same restriction in front of the record loop
    t2.close()
