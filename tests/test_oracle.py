"""CPU suite part 1: the oracle is pinned before it is trusted.

  * against the reference's known-answer vectors (crypto_test.cpp:92-99; vector_smoke tests),
  * against golden fixtures generated from the reference's own translation units
    (tests/golden/make_golden.py), which travel to the GPU box,
  * live against oracle/_ref when it is present (dev container).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import _cases
import _oracle


# ---- SHA-256 ------------------------------------------------------------------------------------
def test_sha256_reference_known_answers(oracle):
    # TestVectors, tests/unit/crypto/crypto_test.cpp:92-99 (reference)
    assert oracle.sha256_hex(b"") == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    assert oracle.sha256_hex(b"abc") == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert oracle.sha256_hex(b"Hello World") == "a591a6d40bf420404a011733cfb7b190d62c65bf0bcda32b57b277d9ad9f146e"


def test_sha256_golden(oracle):
    for c in _cases.load_golden("sha256.json"):
        if "ascii" in c:
            data = c["ascii"].encode()
        else:
            data = np.random.default_rng(c["seed"]).integers(0, 256, c["n"], dtype=np.uint8)
        assert oracle.sha256_hex(data) == c["hex"], c


def test_sha256_vs_hashlib_and_ref(oracle):
    import _oracle
    r = _oracle.ref()
    rng = np.random.default_rng(5)
    for n in list(range(0, 200)) + [4095, 4096, 4097, 65537, 1 << 20]:
        d = rng.integers(0, 256, n, dtype=np.uint8)
        h = oracle.sha256_hex(d)
        assert h == hashlib.sha256(d.tobytes()).hexdigest()
        if r is not None and n < 70000:
            assert h == r.sha256_hex(d)


def test_sha256_split_updates_match_single(ref):
    # "Chunked hashing matches single-pass hashing", crypto_test.cpp:209-228 (reference)
    d = np.random.default_rng(6).integers(0, 256, 1000, dtype=np.uint8).tobytes()
    assert ref.sha256_hex_split(d, [100, 500]) == ref.sha256_hex(d)


# ---- CDC ----------------------------------------------------------------------------------------
def test_rabin_table_default(oracle):
    t = oracle.rabin_table(0x3DA3358B4DC173)
    assert t[0] == 0 and t[1] == 0x3DA3358B4DC173 and t[3] == (0x3DA3358B4DC173 ^ (0x3DA3358B4DC173 << 1))
    # the low 13 bits are injective over byte values
    assert len(set(int(x) & 0x1FFF for x in t)) == 256


def test_cdc_golden(oracle):
    for c in _cases.load_golden("cdc.json"):
        data = _cases.gen_input(c["input"], oracle)
        off, sz = oracle.chunks(data, c["mode"], **c["config"])
        assert [int(x) for x in off] == c["offsets"], (c["input"], c["config"], c["mode"])
        assert [int(x) for x in sz] == c["sizes"]
        hashes = [oracle.sha256_hex(data[int(o):int(o + s)]) for o, s in zip(off, sz)]
        assert hashes[:3] == c["hash_head"] and hashes[-3:] == c["hash_tail"]
        assert oracle.sha256_hex("".join(hashes).encode()) == c["hash_of_hashes"]


def test_cdc_live_vs_reference(oracle, ref):
    rng = np.random.default_rng(21)
    data = rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    for mode in ("rabin", "streaming"):
        for cfg in [{}, {"min_size": 4096, "max_size": 65536}, {"min_size": 1, "max_size": 64, "mask": 7, "window": 5},
                    {"min_size": 100, "max_size": 100}, {"min_size": 300, "max_size": 200, "mask": 0xFF}]:
            d = data if cfg.get("min_size", 16384) >= 4096 else data[:200000]
            a = oracle.chunks(d, mode, **cfg)
            b = ref.chunks(d, mode, with_hashes=False, **cfg)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (mode, cfg)


def test_cdc_invariants(oracle):
    # coverage / contiguity / size bounds, chunking_test.cpp:146-183 (reference)
    data = np.random.default_rng(22).integers(0, 256, 2 << 20, dtype=np.uint8)
    for mode, lo in (("rabin", 4097), ("streaming", 4096)):
        off, sz = oracle.chunks(data, mode, min_size=4096, max_size=32768)
        assert off[0] == 0 and int(off[-1] + sz[-1]) == data.size
        assert np.array_equal(off[1:], (off + sz)[:-1])
        assert (sz[:-1] >= lo).all() and (sz <= 32768).all()


def test_rolling_hash_is_position_local(oracle):
    """the hash at byte n depends only on bytes [n-7..n] and [n-W-7..n-W]; this is
    what lets the device kernel warm up over 8 bytes.  Checked against the sequential ring."""
    rng = np.random.default_rng(23)
    data = rng.integers(0, 256, 5000, dtype=np.uint8)
    W = 48
    t = [int(x) for x in oracle.rabin_table(0x3DA3358B4DC173)]
    M = (1 << 64) - 1
    ring = [0] * W; pos = 0; h = 0; seq = []
    for b in data:
        ob = ring[pos]; ring[pos] = int(b); pos = (pos + 1) % W
        h = (((h - t[ob]) & M) << 8 & M) ^ t[int(b)]
        seq.append(h)
    for n in list(range(0, 120)) + list(rng.integers(120, 5000, 200)):
        g = 0
        for i in range(n - 7, n + 1):
            nb = int(data[i]) if i >= 0 else 0
            ob = int(data[i - W]) if i - W >= 0 else 0
            g = (((g - t[ob]) & M) << 8 & M) ^ t[nb]
        assert g == seq[n], n


# ---- exact scan ---------------------------------------------------------------------------------
def test_scan_reference_known_answers(oracle):
    # "exact scan is a global vector reference", vector_smoke_catch2_test.cpp:188-225 (reference)
    c = np.array([[1.0, i, 0, 0] for i in range(6)], np.float32)
    rows, sims, visited, evals = oracle.scan_cosine(c, np.array([1, 0, 0, 0], np.float32), 3, -1.0)
    assert rows[0] == 0 and len(rows) == 3 and visited == 6 and evals == 6
    # "rejects invalid query vectors" :227-261
    assert oracle.scan_cosine(c, np.zeros(4, np.float32), 1, -1.0) is None
    assert oracle.scan_cosine(c, np.array([1, np.nan, 0, 0], np.float32), 1, -1.0) is None
    # "keeps large finite scores consistent" :263-302
    L = np.float32(np.finfo(np.float32).max / 4)
    e = np.array([[L, -L, L, -L]], np.float32)
    rows, sims, _, _ = oracle.scan_cosine(e, e[0], 1, -1.0)
    assert len(rows) == 1 and np.isfinite(sims[0]) and sims[0] > 0.999
    # "breaks score ties deterministically" :304-353 — ties resolved by chunk_id, not insertion order
    for ids in (["tie_c", "tie_a", "tie_b"], ["tie_b", "tie_a", "tie_c"]):
        rank, _ = _cases.string_ranks(ids)
        c3 = np.tile(np.array([1, 0, 0, 0], np.float32), (3, 1))
        rows, _, _, _ = oracle.scan_cosine(c3, np.array([1, 0, 0, 0], np.float32), 2, -1.0, tie_rank=rank)
        assert [ids[r] for r in rows] == ["tie_a", "tie_b"]


def test_scan_fixture_recipe_cases(oracle):
    # sqlite_vec_backend_comprehensive_catch2_test.cpp:815-844 (reference): 10 seeded vectors dim 64,
    # query = seed 1 -> top-1 is the vector with seed 1 ("chunk_search_0"), 5 results
    corpus = np.stack([_cases.fixture_embedding(64, s + 1) for s in range(10)])
    rows, sims, _, _ = oracle.scan_cosine(corpus, _cases.fixture_embedding(64, 1), 5, 0.0)
    assert rows[0] == 0 and abs(sims[0] - 1.0) < 1e-6
    # :1451-1478 k=100 over 3 rows -> 1..3 results; :1152-1181 thresholds
    rows, _, _, _ = oracle.scan_cosine(corpus[:3], corpus[0], 100, 0.0)
    assert 1 <= len(rows) <= 3
    hi, _, _, _ = oracle.scan_cosine(corpus, corpus[0], 10, 0.99)
    lo, _, _, _ = oracle.scan_cosine(corpus, corpus[0], 10, -1.0)
    assert len(hi) <= 2 and len(lo) == 10


def test_scan_skips_and_cosine_helper(oracle):
    c = np.array([[1, 0, 0, 0], [0, 0, 0, 0], [np.nan, 1, 0, 0], [1e-7, 0, 0, 0], [2, 0, 0, 0]], np.float32)
    rows, sims, visited, evals = oracle.scan_cosine(c, np.array([1, 0, 0, 0], np.float32), 10, -1.0)
    # zero-norm (<=1e-12), NaN and tiny-norm rows are skipped (:4258-4269); rows 0 and 4 tie at 1.0
    assert list(rows) == [0, 4] and visited == 5 and evals == 5
    assert oracle.cosine(c[0], c[4]) == 1.0 and oracle.cosine(c[0], c[1]) == 0.0


def test_scan_record_path(oracle):
    """The metadata-filter path (sqlite_vec_backend.cpp:4333-4409) restated: same scores as the
    fast path, a different zero-norm rule (1e-10 vs 1e-12), full sort, optional AllMatching."""
    # "breaks score ties deterministically" :304-353, the useMetadataFilter = true arm
    for ids in (["tie_c", "tie_a", "tie_b"], ["tie_b", "tie_a", "tie_c"]):
        rank, _ = _cases.string_ranks(ids)
        c3 = np.tile(np.array([1, 0, 0, 0], np.float32), (3, 1))
        rows, _, ev = oracle.scan_cosine_records(c3, np.array([1, 0, 0, 0], np.float32), 2, -1.0, tie_rank=rank)
        assert [ids[r] for r in rows] == ["tie_a", "tie_b"] and ev == 3
    # identical to the fast path when no row norm^2 falls in [1e-12, 1e-10)
    rng = np.random.default_rng(5)
    c = rng.standard_normal((500, 24)).astype(np.float32)
    q = rng.standard_normal(24).astype(np.float32)
    a = oracle.scan_cosine(c, q, 20, 0.1)
    b = oracle.scan_cosine_records(c, q, 20, 0.1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    # a row with norm^2 = 4e-12: kept by the fast path (> 1e-12), dropped by the record path (< 1e-10)
    c2 = np.array([[1, 0, 0, 0], [2e-6, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    q2 = np.array([1, 0, 0, 0], np.float32)
    assert list(oracle.scan_cosine(c2, q2, 3, -1.0)[0]) == [0, 1, 2]
    rows, _, ev = oracle.scan_cosine_records(c2, q2, 3, -1.0)
    assert list(rows) == [0, 2] and ev == 2
    # allow-list (the metadata predicate) and AllMatching (:4398-4400)
    allow = np.zeros(500, np.uint8); allow[::7] = 1
    rows, sims, ev = oracle.scan_cosine_records(c, q, 5, -1.0, allow=allow, all_matching=True)
    assert len(rows) == ev == int(allow.sum()) and (rows % 7 == 0).all() and (np.diff(sims) <= 0).all()


def test_l2_known_answers(oracle):
    # sqlite_vec_c_api_smoke_catch2_test.cpp:20-43 (reference): Euclidean distance WITH sqrt
    a = np.arange(8, dtype=np.float32)
    rows, dist, sims = oracle.scan_l2(np.stack([a, a + 1]), a, 2, -1.0)
    assert list(rows) == [0, 1] and dist[0] == 0.0 and abs(dist[1] - np.sqrt(8)) < 1e-5


def test_mt19937_recipe_matches_the_std_library_golden(oracle):
    """The reference's synthetic-embedding recipe (vector_backend_engine_compare.cpp:83-107):
    the C restatement against a fixture drawn with the real std::mt19937 /
    std::uniform_real_distribution<float> (tests/golden/make_golden.py), and against those classes
    directly when oracle/_ref is present."""
    import hashlib
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mt_recipe.json")))
    sm = g["small"]
    got = oracle.mt19937_rows(sm["seed"], 0, sm["count"], sm["dim"])
    assert [int(x) for x in got.view(np.uint32).ravel()] == sm["values_u32"]
    c1 = g["config1"]
    full = oracle.mt19937_rows(c1["seed"], 0, c1["count"], c1["dim"])
    assert hashlib.sha256(full.tobytes()).hexdigest() == c1["sha256_of_float32_bytes"]
    assert [int(x) for x in full[0, :8].view(np.uint32)] == c1["first_row_head_u32"]
    # queries follow the corpus in the same stream
    q = oracle.mt19937_rows(c1["seed"], 10_000, 16, c1["dim"])
    assert np.array_equal(q.view(np.uint32), full[10_000:].view(np.uint32))
    assert [int(x) for x in q[0, :8].view(np.uint32)] == c1["first_query_head_u32"]
    r = _oracle.ref()
    if r is not None:
        assert np.array_equal(r.mt19937_rows(9, 300, 48).view(np.uint32), oracle.mt19937_rows(9, 0, 300, 48).view(np.uint32))


def test_threaded_scan_equals_one_oracle_call(oracle):
    """_oracle.scan_threaded (slices on all host cores + comparator merge) is the oracle, not an
    approximation of it: identical rows, order and bits, ties across slice boundaries included."""
    n, d, k = 5000, 24, 40
    corpus = oracle.synth_rows(3, 0, n, d)
    corpus[999] = corpus[1000] = corpus[4321] = corpus[17]          # exact ties straddling slices
    q = oracle.synth_rows(3, 1 << 40, 3, d)
    q[0] = corpus[17]
    stats = {}
    got = _oracle.scan_threaded(lambda lo, hi: corpus[lo:hi], n, q, k, slice_rows=1000, threads=4, stats=stats)
    assert stats["slices"] == 5
    for qi in range(3):
        rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, -1.0)
        assert np.array_equal(got[qi][0], rows) and np.array_equal(got[qi][1].view(np.uint32), sims.view(np.uint32))
    got = _oracle.scan_threaded(lambda lo, hi: corpus[lo:hi], n, q, k, metric="l2", thr=0.1, slice_rows=700, threads=3)
    for qi in range(3):
        rows, dist, sims = oracle.scan_l2(corpus, q[qi], k, 0.1)
        assert np.array_equal(got[qi][0], rows) and np.array_equal(got[qi][2].view(np.uint32), dist.view(np.uint32))
        assert np.array_equal(got[qi][1].view(np.uint32), sims.view(np.uint32))


def _hard_corpus(oracle, n, d, seed):
    """Rows that exercise every branch of the scans: exact duplicates (ties broken by row), scaled copies (equal cosine
    up to rounding), zero and near-zero rows (the 1e-12 skip), non-finite elements, huge and tiny magnitudes."""
    c = oracle.synth_rows(seed, 0, n, d)
    rng = np.random.default_rng(seed)
    for _ in range(n // 10):
        i, j = rng.integers(0, n, 2)
        c[i] = c[j]
    for _ in range(n // 20):
        i, j = rng.integers(0, n, 2)
        c[i] = c[j] * np.float32(rng.choice([0.5, 2.0, 3.0, 1e-3]))
    c[rng.integers(0, n, 5)] = 0.0
    c[rng.integers(0, n, 3)] = np.float32(1e-8)
    c[rng.integers(0, n), rng.integers(0, d)] = np.nan
    c[rng.integers(0, n), rng.integers(0, d)] = np.inf
    c[rng.integers(0, n)] *= np.float32(1e18)
    c[rng.integers(0, n)] *= np.float32(1e-18)
    return c


@pytest.mark.parametrize("n,d,nq,k", [(700, 24, 1, 10), (700, 24, 8, 10), (1000, 48, 19, 100), (300, 7, 33, 400), (129, 768, 9, 5),
                                       (64, 16, 8, 1), (65, 16, 7, 64), (0, 16, 3, 4)])
def test_batched_scan_drivers_equal_the_single_query_functions(oracle, n, d, nq, k):
    """oracle_exact_scan_cosine_many / _l2_many (the queries of a batch in the lanes of a vector; what lets a GPU test
    check EVERY query of a 1024-query batch) against oracle_exact_scan_cosine / oracle_exact_scan_l2, the functions the
    reference's known-answer tests pin: rows, order, score bits and counts of every query."""
    corpus = _hard_corpus(oracle, n, d, 11 + n) if n else np.zeros((0, d), np.float32)
    q = oracle.synth_rows(5, 1 << 40, nq, d)
    widths = sorted({oracle.set_lanes(b) for b in (128, 256, 512)})     # every vector width this host offers
    try:
        for w in widths:
            assert oracle.set_lanes(w) == w
            _batched_equals_single(oracle, corpus, q, n, nq, k)
    finally:
        oracle.set_lanes(0)


def _batched_equals_single(oracle, corpus, q, n, nq, k):
    if n:
        ok = [i for i in range(n) if np.isfinite(corpus[i]).all() and 0.01 < float(np.linalg.norm(corpus[i].astype(np.float64))) < 100.0]
        q[0] = corpus[ok[3]]                # a query that IS a row: similarity 1, distance 0, duplicates tie with it
        if nq > 2:
            q[2] = corpus[ok[len(ok) // 2]] * np.float32(4.0)
    for thr in (-1.0, 0.05):
        rows, sims, counts = oracle.scan_cosine_many(corpus, q, k, thr)
        for qi in range(nq):
            r1, s1, _, _ = oracle.scan_cosine(corpus, q[qi], k, thr)
            c = int(counts[qi])
            assert c == len(r1), (qi, c, len(r1))
            assert np.array_equal(rows[qi, :c], r1) and np.array_equal(sims[qi, :c].view(np.uint32), s1.view(np.uint32)), qi
            assert (rows[qi, c:] == -1).all()
    rows, dist, sims, counts = oracle.scan_l2_many(corpus, q, k)
    for qi in range(nq):
        r1, d1, s1 = oracle.scan_l2(corpus, q[qi], k, -1.0)
        c = int(counts[qi])
        assert c == len(r1), (qi, c, len(r1))
        assert np.array_equal(rows[qi, :c], r1) and np.array_equal(dist[qi, :c].view(np.uint32), d1.view(np.uint32)), qi
        assert np.array_equal(sims[qi, :c].view(np.uint32), s1.view(np.uint32)), qi
    # an invalid query fails the batched call as the single call fails (:4127-4130)
    if n:
        bad = q.copy(); bad[nq - 1] = 0.0
        assert oracle.scan_cosine_many(corpus, bad, k) is None and oracle.scan_cosine(corpus, bad[nq - 1], k) is None


def test_threaded_scan_takes_the_batched_drivers_for_many_queries(oracle):
    """scan_threaded with >= MANY_FROM queries (the path of the full-batch checks) equals the single-query oracle."""
    n, d, k, nq = 4000, 32, 30, 21
    corpus = _hard_corpus(oracle, n, d, 77)
    q = oracle.synth_rows(8, 1 << 40, nq, d)
    q[1] = corpus[5]
    assert nq >= _oracle.MANY_FROM
    got = _oracle.scan_threaded(lambda lo, hi: corpus[lo:hi], n, q, k, slice_rows=900, threads=3)
    for qi in range(nq):
        rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, -1.0)
        assert np.array_equal(got[qi][0], rows) and np.array_equal(got[qi][1].view(np.uint32), sims.view(np.uint32))
    got = _oracle.scan_threaded(lambda lo, hi: corpus[lo:hi], n, q, k, metric="l2", thr=0.02, slice_rows=650, threads=3)
    for qi in range(nq):
        rows, dist, sims = oracle.scan_l2(corpus, q[qi], k, 0.02)
        assert np.array_equal(got[qi][0], rows) and np.array_equal(got[qi][2].view(np.uint32), dist.view(np.uint32))
        assert np.array_equal(got[qi][1].view(np.uint32), sims.view(np.uint32))


def test_l2_definitions_fp64_vs_fp32_accumulation_report(oracle, capsys):
    """L2 (BASELINE config 3) is PARITY-UNPINNED: the vec0 arithmetic lives in the absent sqlite-vec-cpp.  This repository
    defines it with fp64 accumulation; the dependency most likely accumulates in fp32.  Distances under both agree well
    inside north_star's 1e-5; index sets may differ at near-ties.  The test REPORTS how often on synthetic rows (it
    asserts only the distance tolerance): the size of the gap, not a parity claim."""
    import _oracle
    n, d, nq, k = 30000, 256, 12, 100
    corpus = oracle.synth_rows(77, 0, n, d)
    q = oracle.synth_rows(77, 1 << 40, nq, d)
    sets = {1: 0, 8: 0, 16: 0}
    worst = 0.0
    for qi in range(nq):
        r64, d64, _ = oracle.scan_l2(corpus, q[qi], k)
        for lanes in sets:
            r32, d32, _ = oracle.scan_l2_f32acc(corpus, q[qi], k, lanes=lanes)
            sets[lanes] += int(set(r32.tolist()) != set(r64.tolist()))
            # same rows or not, the k-th distances agree to fp32 rounding
            worst = max(worst, float(np.abs(d32.astype(np.float64) - d64.astype(np.float64)).max() / d64.max()))
    assert worst < 1e-5
    with capsys.disabled():
        print(f"\n[L2 definition report] {nq} queries x {n} rows x {d}: top-{k} sets that differ from the fp64 definition — "
              f"f32 sequential {sets[1]}, f32 8 lanes {sets[8]}, f32 16 lanes {sets[16]}; worst relative distance difference {worst:.2e}")


def test_two_pass_rank_select_is_the_rank_th_largest_key():
    """The arithmetic of tau_select_kernel's two-pass form (scan_kernels.hip), restated: the rank-th largest of the 256
    per-thread maxima is a lower bound L of the rank-th largest key; the rank-th largest of the keys >= L is the
    answer — including ties, keys of 0 and shares of unequal length."""
    import numpy as np
    rng = np.random.default_rng(7)
    for n_groups, rank, hi in [(12207, 16, 1 << 32), (300, 256, 1 << 32), (20, 16, 1 << 32), (5000, 64, 7), (257, 1, 3), (4096, 33, 1 << 20)]:
        for _ in range(20):
            keys = rng.integers(0, hi, n_groups, dtype=np.uint64).astype(np.uint32)
            want = np.sort(keys)[::-1][rank - 1]
            maxima = np.array([keys[t::256].max() if t < n_groups else 0 for t in range(256)], dtype=np.uint32)
            L = np.sort(maxima)[::-1][rank - 1]
            coll = keys[keys >= L]
            assert len(coll) >= rank
            got = np.sort(coll)[::-1][rank - 1]
            assert got == want and L <= want
