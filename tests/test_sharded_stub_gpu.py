"""The sharded search's kRccl code path with MORE THAN ONE RANK on a one-GPU box.

RCCL refuses two ranks on one device, so until an 8-GPU node runs this library the all-gather branch of
sharded_api.cpp (turnstile, per-rank worker threads calling the collective, empty-record participation of a failed
rank, the exchange fence, destroy with batches in flight) could only ever run with a communicator of ONE rank.  These
tests bind tests/stub_coll — five entry points with RCCL's signatures, stream-ordered copies between "ranks" that
share device 0, ranks that disagree on a collective's size get an error — through the SAME dlopen path the product
uses for librccl.so.1 (`yams_scan_sharded_options_t.rccl_library`), and check every result against the oracle.
Reference call served: SqliteVecBackend::searchSimilarBatch, src/vector/sqlite_vec_backend.cpp:1612-1647."""
import ctypes as C
import random
import threading
import time

import numpy as np
import pytest

from yams_amd import _lib
from yams_amd._lib import SCAN_COSINE, SCAN_L2

pytestmark = pytest.mark.gpu

STUB_VERSION = 9900001


@pytest.fixture(scope="module")
def stub():
    import _cpp_build
    return _cpp_build.build_stub_collective()


def _handle(stub, ranks, lanes, **kw):
    from yams_amd.accel import ShardedScan
    sh = ShardedScan([0] * ranks, lanes=lanes, collective="rccl", rccl_library=stub, **kw)
    info = sh.info()
    assert info["collective"] == "rccl" and info["communicator_ranks"] == ranks, info
    assert info["rccl_version"] == STUB_VERSION and "stub_coll" in info["rccl_library"], info
    assert info["fenced"] is (ranks >= 2 and kw.get("fence", True)), info
    return sh


def _views(sh, corpus, d, ranks, shadows=True):
    """`corpus` as `ranks` contiguous shards uploaded through the handle's contexts (int8 + bf16 shadows)."""
    n = corpus.shape[0]
    keep, views = [], []
    for i in range(ranks):
        lo, hi = n * i // ranks, n * (i + 1) // ranks
        a = sh.ctx(i)
        dc = a.to_device(np.ascontiguousarray(corpus[lo:hi]))
        kw = {}
        if shadows:
            db, dn = a.alloc((hi - lo) * d * 2), a.alloc((hi - lo) * 4)
            a.build_shadow_device(dc.ptr, hi - lo, d, db.ptr, dn.ptr)
            d8, dm8 = a.alloc(_lib.i8_shadow_rows(hi - lo) * d), a.alloc((hi - lo + 15) // 16 * 8)
            a.build_shadow_i8_device(dc.ptr, hi - lo, d, d8.ptr, dm8.ptr)
            keep += [db, dn, d8, dm8]
            kw = dict(rows_bf16_ptr=db.ptr, rows_nsq_ptr=dn.ptr, rows_i8_ptr=d8.ptr, rows_i8_meta_ptr=dm8.ptr)
        a.synchronize()                                 # (the lanes search on their own streams)
        keep.append(dc)
        views.append(a.corpus_view(dc.ptr, hi - lo, d, row_base=lo, **kw))
    return keep, views


def _check(oracle, corpus, q, r, k, thr, metric):
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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("ranks,lanes", [(2, 2), (4, 3), (8, 2), (8, 4)])
def test_all_gather_path_with_several_ranks_equals_the_oracle(oracle, stub, ranks, lanes):
    """2, 4 and 8 ranks x 2-4 lanes, both metrics, int8-tier and narrow batches in flight together: every merged
    result equals the oracle over the whole corpus, one collective per batch; a failing batch (NaN query: every rank
    refuses it) between two good ones fails as a whole and leaves the handle usable."""
    n, d, k = 8 * 5000 + 37, 256, 20
    corpus = oracle.synth_rows(71, 0, n, d)
    corpus[11] = corpus[n - 5] = corpus[n // 2 + 1]      # exact ties across shards
    sh = _handle(stub, ranks, lanes)
    keep, views = _views(sh, corpus, d, ranks)
    batches = [(oracle.synth_rows(71, (1 << 40) + 1000 * j, nq, d), metric, thr)
               for j, (nq, metric, thr) in enumerate([(3, SCAN_COSINE, -1.0), (140, SCAN_COSINE, -1.0), (9, SCAN_L2, 0.05),
                                                      (133, SCAN_L2, -1.0), (1, SCAN_COSINE, 0.02), (17, SCAN_COSINE, -1.0)])]
    batches[0][0][0] = corpus[11]
    done = 0
    pending = []
    for q, metric, thr in batches:
        if len(pending) == lanes:
            lane, (pq, pm, pt) = pending.pop(0)
            _check(oracle, corpus, pq, sh.wait(lane), k, pt, pm); done += 1
        pending.append((sh.submit(views, q, k, thr, metric), (q, metric, thr)))
    for lane, (pq, pm, pt) in pending:
        _check(oracle, corpus, pq, sh.wait(lane), k, pt, pm); done += 1
    assert done == len(batches)
    # a failing batch between two good ones
    bad = batches[0][0].copy(); bad[2, 7] = np.nan
    l1 = sh.submit(views, bad, k, -1.0, SCAN_COSINE)
    l2 = sh.submit(views, batches[5][0], k, -1.0, SCAN_COSINE)
    with pytest.raises(_lib.AccelError) as e:
        sh.wait(l1)
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    _check(oracle, corpus, batches[5][0], sh.wait(l2), k, -1.0, SCAN_COSINE)
    info = sh.info()
    assert info["batches"] == len(batches) + 2 and info["collectives"] == info["batches"], info
    sh.close()


@pytest.mark.timeout(600)
def test_one_rank_that_fails_still_takes_part_in_its_collective(oracle, stub):
    """ONE rank of four refuses its part (its view is malformed: a bf16 shadow without the norms) while the others
    succeed: that rank contributes an empty record, the batch fails as a whole with that rank's error, the collective
    count stays in step on every rank and the next batch — good views again — is oracle-exact.  Then the collective
    itself fails on every rank (stub fault injection): the batch reports an internal error, the handle goes on."""
    ranks, n, d, k = 4, 4 * 4200, 256, 10
    corpus = oracle.synth_rows(72, 0, n, d)
    q = oracle.synth_rows(72, 1 << 40, 6, d)
    sh = _handle(stub, ranks, 2)
    keep, views = _views(sh, corpus, d, ranks)
    broken = list(views)
    v = _lib.ScanCorpus.from_buffer_copy(views[2]); v.rows_nsq = None      # rows_bf16 without rows_nsq -> INVALID_ARG on rank 2 only
    broken[2] = v
    la = sh.submit(views, q, k, -1.0, SCAN_COSINE)
    lb = sh.submit(broken, q, k, -1.0, SCAN_COSINE)
    _check(oracle, corpus, q, sh.wait(la), k, -1.0, SCAN_COSINE)
    with pytest.raises(_lib.AccelError) as e:
        sh.wait(lb)
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG and "rows_nsq" in str(e.value), str(e.value)
    for _ in range(3):
        _check(oracle, corpus, q, sh.topk(views, q, k, -1.0, SCAN_COSINE), k, -1.0, SCAN_COSINE)
    info = sh.info()
    assert info["batches"] == 5 and info["collectives"] == 5, info
    sh.close()


@pytest.mark.timeout(600)
def test_a_collective_that_fails_on_every_rank_fails_its_batch_only(oracle, stub, monkeypatch):
    ranks, n, d, k = 4, 4 * 4200, 256, 10
    corpus = oracle.synth_rows(73, 0, n, d)
    q = oracle.synth_rows(73, 1 << 40, 5, d)
    monkeypatch.setenv("YAMS_STUB_COLL_FAIL_AT", "2")   # (read by the stub when the communicator is formed)
    sh = _handle(stub, ranks, 2)
    monkeypatch.delenv("YAMS_STUB_COLL_FAIL_AT")
    keep, views = _views(sh, corpus, d, ranks)
    for j in range(5):
        if j == 2:
            with pytest.raises(_lib.AccelError) as e:
                sh.topk(views, q, k, -1.0, SCAN_COSINE)
            assert e.value.status == _lib.YAMS_ERR_INTERNAL and "ncclAllGather failed" in str(e.value), str(e.value)
        else:
            _check(oracle, corpus, q, sh.topk(views, q, k, -1.0, SCAN_COSINE), k, -1.0, SCAN_COSINE)
    sh.close()


@pytest.mark.timeout(600)
def test_destroy_with_batches_in_flight(oracle, stub):
    """Every lane holds a submitted batch nobody waits for; destroy lets them run to their end (their collectives need
    every rank) and returns.  A second handle on the same library works afterwards."""
    ranks, n, d, k = 4, 4 * 6000, 256, 10
    corpus = oracle.synth_rows(74, 0, n, d)
    q = oracle.synth_rows(74, 1 << 40, 150, d)
    for _ in range(2):
        sh = _handle(stub, ranks, 3)
        keep, views = _views(sh, corpus, d, ranks)
        for _j in range(3):
            sh.submit(views, q, k, -1.0, SCAN_COSINE)
        t0 = time.time()
        sh.close()
        assert time.time() - t0 < 60
        del keep
    sh = _handle(stub, ranks, 2)
    keep, views = _views(sh, corpus, d, ranks)
    _check(oracle, corpus, q[:4], sh.topk(views, q[:4], k, -1.0, SCAN_COSINE), k, -1.0, SCAN_COSINE)
    sh.close()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("fence", [True, False])
def test_two_hundred_batches_in_random_completion_order(oracle, stub, fence):
    """Four host threads drive the four lanes of an eight-rank handle: 200 batches of 1-160 queries (so the records of
    neighbouring batches differ in size: a collective that paired the wrong batches would be refused by the stub),
    both metrics, random pauses between submit and wait — lanes finish in every order.  Every result equals the
    result of the same batch on ONE context over the whole corpus, which is oracle-checked for a sample of queries of
    every batch; collectives == batches."""
    from yams_amd.accel import Accel
    ranks, lanes, n, d, k = 8, 4, 8 * 4300 + 11, 256, 12
    corpus = oracle.synth_rows(75, 0, n, d)
    sh = _handle(stub, ranks, lanes, fence=fence)
    keep, views = _views(sh, corpus, d, ranks)
    one = Accel(0)
    dc = one.to_device(corpus)
    whole = one.corpus_view(dc.ptr, n, d)
    rng = random.Random(5)
    jobs = []
    for j in range(200):
        nq = rng.choice([1, 2, 3, 5, 8, 13, 31, 64, 129, 160])
        jobs.append((j, oracle.synth_rows(75, (1 << 40) + 1000 * j, nq, d), rng.choice([SCAN_COSINE, SCAN_L2])))
    expect = {}
    for j, q, metric in jobs:
        r = one.scan_topk(whole, q, k, -1.0, metric)
        for qi in sorted({0, q.shape[0] // 2, q.shape[0] - 1}):
            _check(oracle, corpus, q[qi:qi + 1], type(r)(r.scores[qi:qi + 1], r.rows[qi:qi + 1], r.counts[qi:qi + 1], r.dist[qi:qi + 1], r.diag),
                   k, -1.0, metric)
        expect[j] = r
    errors, lock, nxt = [], threading.Lock(), [0]

    def driver(seed):
        r = random.Random(seed)
        try:
            while True:
                with lock:
                    if nxt[0] >= len(jobs):
                        return
                    j, q, metric = jobs[nxt[0]]; nxt[0] += 1
                    lane = sh.submit(views, q, k, -1.0, metric)      # (submit order == the order jobs are drawn)
                time.sleep(r.random() * 0.004)
                got = sh.wait(lane)
                e = expect[j]
                assert np.array_equal(got.counts, e.counts) and np.array_equal(got.rows, e.rows), j
                assert np.array_equal(got.scores.view(np.uint32), e.scores.view(np.uint32)), j
                if metric == SCAN_L2:
                    assert np.array_equal(got.dist.view(np.uint32), e.dist.view(np.uint32)), j
        except BaseException as ex:  # noqa: BLE001 (reported by the main thread)
            errors.append(ex)

    threads = [threading.Thread(target=driver, args=(100 + t,)) for t in range(lanes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]
    info = sh.info()
    assert info["batches"] == 200 and info["collectives"] == 200, info
    print("\n[sharded stub, 8 ranks x 4 lanes, fence=%s] %s" % (fence, info))
    sh.close()
    one.close()


@pytest.mark.timeout(600)
def test_plugin_door_over_the_all_gather_path(accel_lib, oracle, stub):
    """vector_scan_v1 with {"devices": [0,0,0,0], "collective": "rccl", "rccl_library": <stub>}: the mirror is dealt to
    four shards in stripes and every search_batch runs scan -> all-gather -> merge; answers equal the oracle."""
    L = accel_lib
    L.yams_plugin_shutdown()
    cfg = ('{"devices": [0, 0, 0, 0], "stripe_rows": 4096, "search_slots": 3, "collective": "rccl", "rccl_library": "%s"}' % stub).encode()
    assert L.yams_plugin_init(cfg, None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 1, C.byref(p)) == 0
    vt = C.cast(p, C.POINTER(_lib.VectorScanV1)).contents
    n, d, k = 50_003, 256, 15
    corpus = oracle.synth_rows(76, 0, n, d)
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    assert vt.corpus_append(None, cid, corpus.ctypes.data_as(_lib.f32p), n) == 0
    q = oracle.synth_rows(76, 1 << 40, 131, d)
    for nq, metric in ((2, 0), (131, 0), (131, 1)):
        hits = C.POINTER(_lib.ScanHit)(); counts = _lib.u32p(); diag = _lib.ScanDiag()
        qq = np.ascontiguousarray(q[:nq])
        assert vt.search_batch_ex(None, cid, qq.ctypes.data_as(_lib.f32p), nq, d, k, -1.0, metric, 0, None,
                                  C.byref(hits), C.byref(counts), C.byref(diag)) == 0
        for qi in (0, nq // 2, nq - 1):
            rows = (oracle.scan_cosine(corpus, q[qi], k, -1.0) if metric == 0 else oracle.scan_l2(corpus, q[qi], k, -1.0))[0]
            assert [hits[qi * k + i].row for i in range(counts[qi])] == list(rows), (nq, metric, qi)
        vt.free_hits(None, hits, counts)
    info = C.c_void_p()
    assert L.yams_plugin_get_health_json(C.byref(info)) == 0
    js = C.string_at(info).decode()
    C.CDLL(None).free(info)
    import json
    shard_info = json.loads(js)["sharded"]
    assert shard_info["collective"] == "rccl" and "stub_coll" in shard_info["rccl_library"] and shard_info["collectives"] >= 3, js
    assert vt.corpus_destroy(None, cid) == 0
    L.yams_plugin_shutdown()


@pytest.mark.timeout(600)
def test_exchange_fields_of_the_info_record(oracle, stub):
    """What the N > 1 bench line reports about the exchange comes from here: the communicator's rank count as the library
    itself counts it (ncclCommCount), one timed exchange per batch (events on the root shard's side stream around
    all-gather + merge + download), collectives == batches."""
    ranks, n, d, k = 4, 4 * 5000, 256, 10
    corpus = oracle.synth_rows(81, 0, n, d)
    q = oracle.synth_rows(81, 1 << 40, 40, d)
    sh = _handle(stub, ranks, 2)
    keep, views = _views(sh, corpus, d, ranks)
    for _ in range(6):
        _check(oracle, corpus, q[:3], sh.topk(views, q[:3], k, -1.0, SCAN_COSINE), k, -1.0, SCAN_COSINE)
    info = sh.info()
    assert info["communicator_ranks"] == ranks and info["communicator_ranks_source"] == "ncclCommCount", info
    assert info["batches"] == info["collectives"] == info["exchanges_timed"] == 6, info
    assert 0.0 < info["exchange_ms"] <= info["exchange_ms_max"] < 1000.0, info
    assert info["exchange_timeout_ms"] == 30000 and info["stuck"] is False, info
    sh.close()


@pytest.mark.timeout(600)
def test_a_collective_that_never_completes_trips_the_deadline(oracle, stub, monkeypatch):
    """Collective number 2 never completes on any rank's stream (the stub parks a host function there, as an RCCL kernel
    spins when a peer never joins).  wait() must come back within the deadline with YAMS_ERR_TIMEOUT and a diagnosis,
    not hang; the handle is stuck (submit refuses at once), destroy returns promptly (the communicator was aborted,
    the parked streams drain), and a fresh handle on the same devices serves oracle-exact results afterwards."""
    ranks, n, d, k = 4, 4 * 4200, 256, 10
    corpus = oracle.synth_rows(82, 0, n, d)
    q = oracle.synth_rows(82, 1 << 40, 5, d)
    monkeypatch.setenv("YAMS_STUB_COLL_STALL_AT", "2")
    sh = _handle(stub, ranks, 2, exchange_timeout_ms=1500)
    monkeypatch.delenv("YAMS_STUB_COLL_STALL_AT")
    keep, views = _views(sh, corpus, d, ranks)
    for _ in range(2):
        _check(oracle, corpus, q, sh.topk(views, q, k, -1.0, SCAN_COSINE), k, -1.0, SCAN_COSINE)
    t0 = time.time()
    with pytest.raises(_lib.AccelError) as e:
        sh.topk(views, q, k, -1.0, SCAN_COSINE)
    took = time.time() - t0
    assert e.value.status == _lib.YAMS_ERR_TIMEOUT, str(e.value)
    assert 1.4 < took < 20.0, took
    msg = str(e.value)
    assert "batch 2" in msg and "1500 ms" in msg and "4 ranks" in msg, msg
    info = sh.info()
    assert info["stuck"] is True and info["exchanges_timed"] == 2, info
    with pytest.raises(_lib.AccelError) as e2:
        sh.topk(views, q, k, -1.0, SCAN_COSINE)
    assert e2.value.status == _lib.YAMS_ERR_TIMEOUT and "stuck" in str(e2.value), str(e2.value)
    t0 = time.time()
    sh.close()
    assert time.time() - t0 < 30.0
    del keep
    sh = _handle(stub, ranks, 2)
    keep, views = _views(sh, corpus, d, ranks)
    _check(oracle, corpus, q, sh.topk(views, q, k, -1.0, SCAN_COSINE), k, -1.0, SCAN_COSINE)
    sh.close()
