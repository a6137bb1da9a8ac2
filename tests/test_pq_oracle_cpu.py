"""oracle_pq_search (oracle/yams_oracle.c) — the restatement of SqliteVecBackend::Impl::simeonPqSearchUnlocked
(src/vector/sqlite_vec_backend.cpp:3868-4056) — against a second, independent reading of the same text in numpy
(tests/_pq.py).  PARITY UNPINNED: third_party/simeon is absent from the reference checkout, so neither the order of the
ADC sum nor the quantiser itself can be compared with the reference's; what these tests hold is that the oracle does what
its header says, on every branch the reference's function has."""
import numpy as np
import pytest

import _pq


def case(oracle, n=600, d=48, m=8, seed=3, dup=True):
    corpus = oracle.synth_rows(seed, 0, n, d) * np.linspace(0.5, 2.0, n, dtype=np.float32)[:, None]   # raw rows: not unit
    if dup:
        corpus[40:48] = corpus[7]            # equal exact similarities AND equal codes
    pq = _pq.Pq(_pq.unit(corpus), m, seed)
    codes = pq.encode(_pq.unit(corpus))
    ids = ["c%05d" % ((97 * i) % 1009) for i in range(n)]
    keys = np.array([_pq.stable_string_key(s) for s in ids], np.uint64)
    rank = np.argsort(np.argsort(np.array(ids))).astype(np.uint64)
    return corpus, pq, codes, keys, rank


@pytest.mark.parametrize("lanes", [1, 4, 8, 16])
def test_oracle_pq_search_equals_the_numpy_reading(oracle, lanes):
    corpus, pq, codes, keys, rank = case(oracle)
    queries = oracle.synth_rows(3, 1 << 40, 4, corpus.shape[1]) * np.float32(3.0)
    for qi, q in enumerate(queries):
        lut = pq.lut(q)
        for k, rf, thr in ((5, 2, -1.0), (10, 1, -1.0), (10, 4, 0.2), (700, 2, -1.0), (3, 2, 0.9)):
            rows, sims, st = oracle.pq_search(corpus, codes, lut, q, k, thr, rf, tie_keys=keys, chunk_rank=rank, sum_lanes=lanes)
            nr, ns = _pq.numpy_pq_search(corpus, codes, lut, q, k, thr, rf, keys, None, rank, None, lanes)
            assert rows.tolist() == nr, (qi, k, rf, thr)
            assert np.array_equal(sims.view(np.uint32), np.array(ns, np.float32).view(np.uint32))
            assert st["candidates"] == corpus.shape[0] and st["materialised"] == min(corpus.shape[0], max(k, k * rf))


def test_oracle_pq_search_candidates_missing_rows_and_refused_queries(oracle):
    corpus, pq, codes, keys, rank = case(oracle, n=400)
    n, d = corpus.shape
    q = oracle.synth_rows(3, 1 << 41, 1, d)[0]
    lut = pq.lut(q)
    cand = np.arange(3, n, 5, dtype=np.uint32)
    roi = np.arange(n, dtype=np.uint32)
    roi[[8, 13, 18]] = n + 7                      # indexed rows the vectors table has lost (:4010-4012)
    rows, sims, st = oracle.pq_search(corpus, codes, lut, q, 12, -1.0, 3, tie_keys=keys, row_of_index=roi, chunk_rank=rank, candidates=cand)
    nr, ns = _pq.numpy_pq_search(corpus, codes, lut, q, 12, -1.0, 3, keys, roi, rank, cand, 1)
    assert rows.tolist() == nr and st["candidates"] == cand.size
    assert set(rows.tolist()) <= set(cand.tolist()) and not (set(rows.tolist()) & {8, 13, 18})
    # an empty candidate list, k == 0, and a query the host's normalisation refuses return nothing
    assert len(oracle.pq_search(corpus, codes, lut, q, 5, candidates=np.zeros(0, np.uint32))[0]) == 0
    assert len(oracle.pq_search(corpus, codes, lut, q, 0)[0]) == 0
    assert len(oracle.pq_search(corpus, codes, lut, np.zeros(d, np.float32), 5)[0]) == 0
    assert len(oracle.pq_search(corpus, codes, lut, np.full(d, 1e-12, np.float32), 5)[0]) == 0
    assert len(oracle.pq_search(corpus, codes, lut, np.full(d, 1e-9, np.float32), 5)[0]) == 5       # norm^2 = 4.8e-17 > 1e-20


def test_the_sum_orders_are_distinguishable(oracle):
    """A crafted table separates the four served orders of the ADC sum (the calibration idea of the L2 path): values that
    cancel differently in a sequential fp32 sum and in 4 / 8 / 16 partial sums."""
    m = 32
    code = np.zeros(m, np.uint8)
    for seed in range(64):
        rng = np.random.default_rng(seed)
        lut = np.zeros((m, 256), np.float32)
        lut[:, 0] = (rng.standard_normal(m) * 10.0 ** rng.integers(-3, 5, m)).astype(np.float32)
        got = {l: oracle.pq_adc_score(code, lut, l) for l in (1, 4, 8, 16)}
        if len({np.float32(v).tobytes() for v in got.values()}) == 4:
            return
    raise AssertionError("no table among 64 separated the four sum orders")
