"""Device memory exhausted — SURVEY.md 5: `ErrorCode::ResourceExhausted` (include/yams/core/types.h:49 of the reference).
A mirror costs 7 bytes per element (fp32 rows + bf16 + int8 shadows: 67 GB for a 12.5M x 768 shard), so exhaustion is a
state a live `vectors.db` reaches.  yams_accel_debug_fail_alloc_after(n) makes the n-th allocation the library performs
for its own objects fail exactly as hipErrorOutOfMemory does; what must hold then:
  * the failing call returns YAMS_ERR_RESOURCE_EXHAUSTED (7), not an internal error;
  * the object it was growing is left as it was: a corpus keeps its rows and answers searches over them, oracle-exact;
  * nothing leaks (health JSON: mirror_bytes_mapped does not move on a failed append);
  * the same call succeeds once memory is there again, and after corpus_clear.
Every result is compared with the oracle; the C++ adapter's side (ErrorCode::ResourceExhausted through AccelVectorIndex)
is tests/cpp/host_mirror_test.cpp."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from yams_amd import _lib

pytestmark = pytest.mark.gpu

# The injection exists in the MEASUREMENT build only (the product library's doors are inert: ADVICE r5): a plain run of this
# file re-runs it once with YAMS_ACCEL_MEASURE_LIB=1, where the tests below do their work.
_MEASURE = bool(os.environ.get("YAMS_ACCEL_MEASURE_LIB"))
needs_injection = pytest.mark.skipif(not _MEASURE, reason="allocation-failure injection is compiled into the measurement build only")


@pytest.mark.skipif(_MEASURE, reason="this IS the measurement-build run")
def test_exhaustion_tests_pass_on_the_measurement_build(accel_lib):
    """The product library cannot be armed (the doors are there — one ABI — and do nothing) ..."""
    import subprocess, sys
    from yams_amd import build as _build
    assert accel_lib.yams_accel_debug_alloc_injection_compiled() == 0
    accel_lib.yams_accel_debug_fail_alloc_after(0)
    assert accel_lib.yams_accel_debug_alloc_faults() == 0
    accel_lib.yams_accel_debug_fail_alloc_after(-1)
    # ... and the exhaustion tests run against the measurement build of the same sources
    if not os.path.exists(_build.MEASURE_LIB):
        _build.build(measure=True)
    env = dict(os.environ, YAMS_ACCEL_MEASURE_LIB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def _health(L):
    p = C.c_void_p()
    assert L.yams_plugin_get_health_json(C.byref(p)) == 0
    try:
        return json.loads(C.string_at(p).decode())
    finally:
        L.yams_accel_free_string(p)


def _vt(L, config=b'{"device": 0}'):
    L.yams_plugin_shutdown()
    assert L.yams_plugin_init(config, None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 1, C.byref(p)) == 0
    return C.cast(p, C.POINTER(_lib.VectorScanV1)).contents


def _search(vt, cid, q, k, expect=0):
    nq, d = q.shape
    hits = C.POINTER(_lib.ScanHit)(); counts = _lib.u32p(); diag = _lib.ScanDiag()
    st = vt.search_batch_ex(None, cid, q.ctypes.data_as(_lib.f32p), nq, d, k, -1.0, 0, 0, None, C.byref(hits), C.byref(counts), C.byref(diag))
    assert st == expect, st
    if st != 0:
        return None
    rows = [[hits[qi * k + i].row for i in range(counts[qi])] for qi in range(nq)]
    sims = [np.array([hits[qi * k + i].similarity for i in range(counts[qi])], np.float32) for qi in range(nq)]
    vt.free_hits(None, hits, counts)
    return rows, sims


def _exact(oracle, corpus, q, got, k):
    for qi in range(q.shape[0]):
        rows, sims, _, _ = oracle.scan_cosine(corpus, q[qi], k, -1.0)
        assert got[0][qi] == list(rows), qi
        assert np.array_equal(got[1][qi].view(np.uint32), sims.view(np.uint32)), qi


@pytest.fixture()
def armed(accel_lib):
    """Injection is process-wide: always disarmed again, whatever the test did."""
    yield accel_lib
    accel_lib.yams_accel_debug_fail_alloc_after(-1)


@needs_injection
@pytest.mark.timeout(600)
@pytest.mark.parametrize("config", [b'{"device": 0}', b'{"devices": [0, 0], "stripe_rows": 4096}'])
def test_an_append_that_exhausts_memory_leaves_the_corpus_serving(armed, oracle, config):
    L = armed
    vt = _vt(L, config)
    d, k = 256, 10
    n0, n1 = 20_000, 400_000                       # 20 MB first, then 400 MB more: the mirrors must map fresh memory
    corpus = oracle.synth_rows(61, 0, n0 + n1, d)
    q = oracle.synth_rows(61, 1 << 40, 5, d)
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    assert vt.corpus_append(None, cid, corpus[:n0].ctypes.data_as(_lib.f32p), n0) == 0
    _exact(oracle, corpus[:n0], q, _search(vt, cid, q, k), k)
    h0 = _health(L)
    faults0 = L.yams_accel_debug_alloc_faults()
    more = np.ascontiguousarray(corpus[n0:])
    L.yams_accel_debug_fail_alloc_after(0)          # the very first allocation of the append fails
    assert vt.corpus_append(None, cid, more.ctypes.data_as(_lib.f32p), n1) == _lib.YAMS_ERR_RESOURCE_EXHAUSTED
    L.yams_accel_debug_fail_alloc_after(-1)
    exhausted, st = 1, None
    for attempt in range(8):                         # one allocation succeeds per attempt, the next one fails: rows, then
        L.yams_accel_debug_fail_alloc_after(1)       # the bf16 shadow, then the int8 shadow get their memory, one array per try
        st = vt.corpus_append(None, cid, more.ctypes.data_as(_lib.f32p), n1)
        L.yams_accel_debug_fail_alloc_after(-1)
        if st == 0:
            break
        assert st == _lib.YAMS_ERR_RESOURCE_EXHAUSTED, (attempt, st)
        exhausted += 1
        nn = C.c_uint64()
        assert vt.corpus_size(None, cid, C.byref(nn), None) == 0 and nn.value == n0      # not one row more
        _exact(oracle, corpus[:n0], q, _search(vt, cid, q, k), k)                         # ... and it still answers
    assert st == 0 and exhausted >= 3, (st, exhausted)      # memory trickled in: the append went through in the end
    _exact(oracle, corpus, q, _search(vt, cid, q, k), k)
    h1 = _health(L)
    assert L.yams_accel_debug_alloc_faults() >= faults0 + exhausted and h1["exhausted_appends"] == h0["exhausted_appends"] + exhausted
    # memory mapped by the failed attempts stayed WITH the corpus and was used by the attempt that succeeded: what is
    # mapped now is what the rows and their shadows need (plus the growth headroom), not a multiple of it
    h2 = h1
    need = (n0 + n1) * d * 7
    assert need <= h2["mirror_bytes_mapped"] <= 2.2 * need + (200 << 20), (h2["mirror_bytes_mapped"], need)
    # corpus_clear, then an exhausted append, then a good one
    assert vt.corpus_clear(None, cid) == 0
    L.yams_accel_debug_fail_alloc_after(0)
    big = oracle.synth_rows(62, 0, 1_200_000, d)     # larger than anything mapped so far
    st = vt.corpus_append(None, cid, big.ctypes.data_as(_lib.f32p), big.shape[0])
    L.yams_accel_debug_fail_alloc_after(-1)
    assert st == _lib.YAMS_ERR_RESOURCE_EXHAUSTED
    assert vt.corpus_append(None, cid, corpus[:n0].ctypes.data_as(_lib.f32p), n0) == 0
    _exact(oracle, corpus[:n0], q, _search(vt, cid, q, k), k)
    assert vt.corpus_destroy(None, cid) == 0
    h3 = _health(L)
    assert h3["mirror_bytes_mapped"] == 0 and h3["mirror_bytes_parked"] > 0            # parked for the next corpus, not leaked
    L.yams_plugin_shutdown()


@needs_injection
@pytest.mark.timeout(600)
def test_a_search_whose_workspace_cannot_grow_fails_alone(armed, oracle):
    """The scan's workspace and the sharded lanes' batch buffers grow with the batch: a batch that cannot get them fails
    with RESOURCE_EXHAUSTED, smaller batches go on being served, and the big one succeeds afterwards."""
    L = armed
    vt = _vt(L, b'{"device": 0, "search_slots": 1}')
    n, d, k = 60_000, 256, 10
    corpus = oracle.synth_rows(63, 0, n, d)
    q = oracle.synth_rows(63, 1 << 40, 300, d)
    cid = C.c_uint64()
    assert vt.corpus_create(None, d, C.byref(cid)) == 0
    assert vt.corpus_append(None, cid, corpus.ctypes.data_as(_lib.f32p), n) == 0
    small = np.ascontiguousarray(q[:4])
    _exact(oracle, corpus, small, _search(vt, cid, small, k), k)
    for after in (0, 2, 5):
        L.yams_accel_debug_fail_alloc_after(after)
        got = _search(vt, cid, q, k, expect=_lib.YAMS_ERR_RESOURCE_EXHAUSTED)      # 300 queries: every buffer must grow
        L.yams_accel_debug_fail_alloc_after(-1)
        assert got is None
        _exact(oracle, corpus, small, _search(vt, cid, small, k), k)
    got = _search(vt, cid, q, k)
    _exact(oracle, corpus, q[:6], (got[0][:6], got[1][:6]), k)
    assert vt.corpus_destroy(None, cid) == 0
    L.yams_plugin_shutdown()


@needs_injection
@pytest.mark.timeout(600)
def test_flat_entry_points_report_exhaustion(armed, oracle):
    """yams_scan_topk (workspace), yams_dedup_set_create (digest set), yams_ingest_device (bitmaps / slots): status 7 with a
    message that names the allocation; the context stays usable."""
    L = armed
    from yams_amd.accel import Accel
    acc = Accel(0)                                      # a context of its own: no workspace grown by an earlier test
    n, d = 30_000, 128
    corpus = oracle.synth_rows(64, 0, n, d)
    q = oracle.synth_rows(64, 1 << 40, 40, d)
    dc = acc.to_device(corpus)
    view = acc.corpus_view(dc.ptr, n, d)
    L.yams_accel_debug_fail_alloc_after(0)
    with pytest.raises(_lib.AccelError) as e:
        acc.scan_topk(view, q, 300, -1.0)               # k = 300: a workspace no earlier test of this context asked for
    L.yams_accel_debug_fail_alloc_after(-1)
    assert e.value.status == _lib.YAMS_ERR_RESOURCE_EXHAUSTED and "hipMalloc workspace" in str(e.value), str(e.value)
    r = acc.scan_topk(view, q[:3], 5, -1.0)
    for qi in range(3):
        assert list(r.rows[qi, :5]) == list(oracle.scan_cosine(corpus, q[qi], 5, -1.0)[0])
    L.yams_accel_debug_fail_alloc_after(0)
    with pytest.raises(_lib.AccelError) as e:
        acc.dedup_set(1 << 20)
    L.yams_accel_debug_fail_alloc_after(-1)
    assert e.value.status == _lib.YAMS_ERR_RESOURCE_EXHAUSTED, str(e.value)
    s = acc.dedup_set(1000)
    s.close()
    acc.close()
