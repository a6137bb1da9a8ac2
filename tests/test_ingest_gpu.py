"""GPU parity suite for the content-ingest path (SHA-256 + content-defined chunking), through the
C ABI, against the oracle and the golden fixtures generated from the reference's own sources.
Bar: bit-exact digests and chunk boundaries."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _cases
from yams_amd import _lib
from yams_amd.accel import cdc_config

pytestmark = pytest.mark.gpu


# ---- SHA-256 ------------------------------------------------------------------------------------
def test_sha256_reference_known_answers(acc):
    # tests/unit/crypto/crypto_test.cpp:92-99,134-170 (reference)
    assert acc.sha256_hex(b"") == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    assert acc.sha256_hex(b"abc") == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert acc.sha256_hex(b"Hello World") == "a591a6d40bf420404a011733cfb7b190d62c65bf0bcda32b57b277d9ad9f146e"


def test_sha256_golden(acc):
    for c in _cases.load_golden("sha256.json"):
        data = c["ascii"].encode() if "ascii" in c else \
            np.random.default_rng(c["seed"]).integers(0, 256, c["n"], dtype=np.uint8)
        assert acc.sha256_hex(data) == c["hex"], c


def test_sha256_every_length_and_alignment(acc, oracle):
    rng = np.random.default_rng(41)
    base = rng.integers(0, 256, 5000, dtype=np.uint8)
    # every length 0..300 (all padding cases) at every byte alignment 0..3 of the source
    msgs = [base[a:a + n] for n in range(0, 301) for a in range(4)]
    msgs += [rng.integers(0, 256, n, dtype=np.uint8) for n in (4095, 4096, 4097, 65537, 1 << 20, (1 << 20) + 3)]
    got = acc.sha256_many(msgs)
    for h, m in zip(got, msgs):
        assert h == hashlib.sha256(m.tobytes()).hexdigest(), len(m)
    assert got[7] == oracle.sha256_hex(msgs[7])


def test_sha256_ragged_batch_on_device(acc):
    """Device-resident batch with unaligned offsets: the lane queue must not lose or mix messages."""
    import torch
    rng = np.random.default_rng(42)
    blob = rng.integers(0, 256, 3_000_000, dtype=np.uint8)
    n = 5000
    lens = rng.integers(0, 1500, n).astype(np.uint64)
    lens[::97] = rng.integers(20000, 60000, len(lens[::97]))
    offs = rng.integers(0, blob.size - 60000, n).astype(np.uint64)
    tb = torch.from_numpy(blob).cuda()
    to, tl = torch.from_numpy(offs.view(np.int64)).cuda(), torch.from_numpy(lens.view(np.int64)).cuda()
    dg = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    acc.sha256_batch_device(tb.data_ptr(), to.data_ptr(), tl.data_ptr(), n, dg.data_ptr())
    acc.synchronize()
    got = dg.cpu().numpy()
    for i in range(n):
        exp = hashlib.sha256(blob[int(offs[i]):int(offs[i] + lens[i])].tobytes()).digest()
        assert got[i].tobytes() == exp, i


def test_sha256_streaming_vtable_matches_one_shot(accel_lib):
    # "Chunked hashing matches single-pass hashing", crypto_test.cpp:209-228; sizes :172-187
    L = accel_lib
    assert L.yams_plugin_init(b"{}", None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"content_hash_v1", 1, C.byref(p)) == 0
    vt = C.cast(p, C.POINTER(_lib.ContentHashV1)).contents
    rng = np.random.default_rng(43)
    st = C.c_void_p()
    assert vt.stream_create(None, C.byref(st)) == 0
    out = C.create_string_buffer(65)
    for size, cuts in [(0, []), (1, []), (17, [5]), (1000, [100, 500]), (4096, [64, 128, 4000]),
                       (65537, [1, 63, 64, 65, 30000])]:
        data = rng.integers(0, 256, size, dtype=np.uint8)
        assert vt.stream_init(None, st) == 0
        prev = 0
        for c in cuts + [size]:
            piece = np.ascontiguousarray(data[prev:c])
            assert vt.stream_update(None, st, piece.ctypes.data_as(_lib.u8p) if piece.size else None, piece.size) == 0
            prev = c
        assert vt.stream_finalize(None, st, out) == 0
        assert out.value.decode() == hashlib.sha256(data.tobytes()).hexdigest(), size
        one = C.create_string_buffer(65)
        assert vt.hash(None, data.ctypes.data_as(_lib.u8p) if size else None, size, one) == 0
        assert one.value == out.value
    # finalize re-initialises the stream (sha256_hasher.cpp:103-106): hashing nothing gives sha256("")
    assert vt.stream_finalize(None, st, out) == 0
    assert out.value.decode() == hashlib.sha256(b"").hexdigest()
    vt.stream_destroy(None, st)
    L.yams_plugin_shutdown()


# ---- content-defined chunking -------------------------------------------------------------------
def _check_chunks(acc, oracle, data, mode, **cfg):
    off, sz, hx = acc.chunk(data, cdc_config(mode, **cfg), with_hashes=True)
    ooff, osz = oracle.chunks(data, mode, **cfg)
    assert len(off) == len(ooff), (mode, cfg, len(off), len(ooff))
    assert np.array_equal(off, ooff) and np.array_equal(sz, osz), (mode, cfg)
    step = max(1, len(off) // 64)
    for i in list(range(0, len(off), step)) + [len(off) - 1] if len(off) else []:
        assert hx[i] == hashlib.sha256(data[int(off[i]):int(off[i] + sz[i])].tobytes()).hexdigest()
    return off, sz, hx


def test_cdc_golden_from_reference_sources(acc, oracle):
    for c in _cases.load_golden("cdc.json"):
        data = _cases.gen_input(c["input"], oracle)
        off, sz, hx = acc.chunk(data, cdc_config(c["mode"], **c["config"]), with_hashes=True)
        assert [int(x) for x in off] == c["offsets"], (c["input"], c["config"], c["mode"])
        assert [int(x) for x in sz] == c["sizes"]
        assert hx[:3] == c["hash_head"] and hx[-3:] == c["hash_tail"]
        assert hashlib.sha256("".join(hx).encode()).hexdigest() == c["hash_of_hashes"]


@pytest.mark.parametrize("mode", ["rabin", "streaming"])
def test_cdc_config_sweep(acc, oracle, mode):
    rng = np.random.default_rng(44)
    data = rng.integers(0, 256, 6 << 20, dtype=np.uint8)
    _check_chunks(acc, oracle, data, mode)                                       # product default
    _check_chunks(acc, oracle, data, mode, min_size=4096, max_size=65536)        # core_benchmarks.cpp:226-227
    small = data[:400000]
    for cfg in [dict(min_size=64, max_size=256, mask=0xF), dict(min_size=1, max_size=100, mask=3, window=16),
                dict(min_size=2048, max_size=8192, mask=0xFFFFF), dict(min_size=512, max_size=4096, mask=(1 << 40) - 1),
                dict(min_size=100, max_size=100), dict(min_size=300, max_size=200, mask=0xFF),
                dict(min_size=1000, max_size=3000, mask=0xFFFFFFFFFFFFFFFF), dict(min_size=37, max_size=4001, mask=0x155, window=1),
                dict(min_size=5000, max_size=9000, mask=0x3FF, window=7, polynomial=0xBFE6B8A5BF378D83),
                dict(min_size=0, max_size=50, mask=1),
                # both sides of the 2^31 mask limit of the narrow (32-bit) candidate kernel, and mask 0
                dict(min_size=8, max_size=5000, mask=0x7FFFFFFF), dict(min_size=8, max_size=5000, mask=0x80000000),
                dict(min_size=8, max_size=5000, mask=0x40000001), dict(min_size=40, max_size=90, mask=0)]:
        _check_chunks(acc, oracle, small, mode, **cfg)


@pytest.mark.parametrize("mode", ["rabin", "streaming"])
def test_cdc_edge_inputs(acc, oracle, mode):
    # "Empty input produces no chunks" (chunking_test.cpp:108-113), sizes around min/max and < window
    off, sz, hx = acc.chunk(np.zeros(0, np.uint8), cdc_config(mode))
    assert len(off) == 0
    rng = np.random.default_rng(45)
    for n in [1, 7, 47, 48, 49, 55, 56, 57, 4095, 4096, 4097, 8191, 8192, 8193, 32767, 32768, 32769, 65600]:
        d = rng.integers(0, 256, n, dtype=np.uint8)
        _check_chunks(acc, oracle, d, mode, min_size=4096, max_size=8192, mask=0x3FF)
    _check_chunks(acc, oracle, np.zeros(3 << 20, np.uint8), mode)               # no candidates: max-size chunks
    _check_chunks(acc, oracle, np.full(200000, 0x42, np.uint8), mode, min_size=4096, max_size=32768)
    _check_chunks(acc, oracle, _cases.pattern(256 * 1024 + 777), mode, min_size=2048, max_size=65536)


def test_cdc_rabin_equals_streaming_on_constant_buffer(acc):
    # chunking_test.cpp:230-251 (reference): identical boundaries on a constant buffer
    d = np.full(1 << 20, 0x42, np.uint8)
    a = acc.chunk(d, cdc_config("rabin", min_size=4096, max_size=65536), with_hashes=False)
    b = acc.chunk(d, cdc_config("streaming", min_size=4096, max_size=65536), with_hashes=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_cdc_invalid_configs_are_rejected(acc):
    d = np.zeros(1000, np.uint8)
    for cfg in [cdc_config("rabin", window=0), cdc_config("rabin", window=49), cdc_config("rabin", min_size=0, max_size=0)]:
        with pytest.raises(_lib.AccelError) as e:
            acc.chunk(d, cfg)
        assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    # the streaming chunker clamps the ring instead (streaming_chunker.cpp:44-49)
    acc.chunk(d, cdc_config("streaming", window=0, min_size=10, max_size=100))
    acc.chunk(d, cdc_config("streaming", window=500, min_size=10, max_size=100))


def test_ingest_many_blobs_unaligned(acc, oracle):
    """The ingest hot path over a ragged blob set placed at arbitrary byte offsets: boundaries,
    per-chunk digests and whole-blob digests, all against the CPU."""
    import torch
    rng = np.random.default_rng(46)
    lens = [0, 1, 47, 5000, 16384, 16385, 100_000, 1 << 20, (1 << 20) + 1, 3_333_333, 40, 70_001]
    gaps = [3, 1, 0, 7, 13, 2, 5, 1, 9, 0, 11, 6]
    blobs = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    offs, pos, parts = [], 5, [np.zeros(5, np.uint8)]
    for b, g in zip(blobs, gaps):
        offs.append(pos); parts.append(b); parts.append(np.zeros(g, np.uint8)); pos += len(b) + g
    buf = np.concatenate(parts + [np.zeros(64, np.uint8)])
    tb = torch.from_numpy(buf).cuda()
    for mode in ("streaming", "rabin"):
        cfg = dict(min_size=4096, max_size=65536)
        res = acc.ingest_device(tb.data_ptr(), offs, lens, cdc_config(mode, **cfg), flags=3)
        out = acc.fetch_ingest(res, len(lens))
        first = out["blob_first"]
        for bi, b in enumerate(blobs):
            ooff, osz = oracle.chunks(b, mode, **cfg)
            lo, hi = int(first[bi]), int(first[bi + 1])
            assert hi - lo == len(ooff), (mode, bi)
            assert np.array_equal(out["chunk_offset"][lo:hi], ooff) and np.array_equal(out["chunk_size"][lo:hi], osz)
            assert (out["chunk_blob"][lo:hi] == bi).all()
            assert out["blob_digest"][bi].tobytes() == hashlib.sha256(b.tobytes()).digest()
            for j in range(lo, hi):
                o, s = int(out["chunk_offset"][j]), int(out["chunk_size"][j])
                assert out["chunk_digest"][j].tobytes() == hashlib.sha256(b[o:o + s].tobytes()).digest()


def test_ingest_full_size_properties(acc, oracle):
    """A slice of BASELINE config 5 (Philox blobs, 4 MiB each, product-default chunker) large
    enough to fill the device: invariants on everything, the CPU oracle on a sample of blobs."""
    import torch
    n_blobs, blen = 256, 4 << 20                      # 1 GiB
    tb = torch.empty(n_blobs * blen, dtype=torch.uint8, device="cuda")
    acc.synth_bytes(42, 0, n_blobs, blen, tb.data_ptr())
    offs = [i * blen for i in range(n_blobs)]
    res = acc.ingest_device(tb.data_ptr(), offs, [blen] * n_blobs, cdc_config("streaming"), flags=3)
    out = acc.fetch_ingest(res, n_blobs)
    first, co, cs = out["blob_first"], out["chunk_offset"], out["chunk_size"]
    assert first[0] == 0 and first[-1] == out["n_chunks"]
    for bi in range(n_blobs):                         # coverage, contiguity, size bounds (chunking_test.cpp:146-183)
        lo, hi = int(first[bi]), int(first[bi + 1])
        o, s = co[lo:hi], cs[lo:hi]
        assert o[0] == 0 and int(o[-1] + s[-1]) == blen
        assert np.array_equal(o[1:], (o + s)[:-1])
        assert (s[:-1] >= 16384).all() and (s <= 1 << 20).all()
    assert 20000 < out["n_chunks"] / n_blobs * 256 < 60000   # avg chunk ~24.5 KB on random data
    for bi in (0, 101, 255):                          # bit-exact against the CPU
        blob = oracle.synth_bytes(42, bi, 0, blen)
        ooff, osz = oracle.chunks(blob, "streaming")
        lo, hi = int(first[bi]), int(first[bi + 1])
        assert np.array_equal(co[lo:hi], ooff) and np.array_equal(cs[lo:hi], osz)
        assert out["blob_digest"][bi].tobytes() == hashlib.sha256(blob.tobytes()).digest()
        for j in range(lo, hi, 7):
            assert out["chunk_digest"][j].tobytes() == \
                hashlib.sha256(blob[int(co[j]):int(co[j] + cs[j])].tobytes()).digest()
    # determinism: the same call again gives the same chunk digests (chunking_test.cpp:185-228)
    res2 = acc.ingest_device(tb.data_ptr(), offs, [blen] * n_blobs, cdc_config("streaming"), flags=3)
    out2 = acc.fetch_ingest(res2, n_blobs)
    assert np.array_equal(out2["chunk_digest"], out["chunk_digest"])


def test_ingest_config5_8GiB_every_blob_verified(acc, oracle):
    """8 GiB of BASELINE config 5 (2048 Philox blobs x 4 MiB, product-default StreamingChunker),
    EVERY blob checked on the host cores: the blob is regenerated from the Philox recipe on the CPU
    (so the device generator is checked too), chunked by the reference's own StreamingChunker
    (oracle/_ref) or its C restatement, and all chunk digests + the blob digest are compared with
    hashlib (OpenSSL, what the reference links)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    import _oracle
    n_blobs, blen = 2048, 4 << 20
    tb = torch.empty(n_blobs * blen, dtype=torch.uint8, device="cuda")
    acc.synth_bytes(42, 0, n_blobs, blen, tb.data_ptr())
    offs = [i * blen for i in range(n_blobs)]
    res = acc.ingest_device(tb.data_ptr(), offs, [blen] * n_blobs, cdc_config("streaming"), flags=3)
    out = acc.fetch_ingest(res, n_blobs)
    first, co, cs = out["blob_first"], out["chunk_offset"], out["chunk_size"]
    cd, bd = out["chunk_digest"], out["blob_digest"]
    ref = _oracle.ref()

    def verify(bi):
        blob = oracle.synth_bytes(42, bi, 0, blen)
        ref_hashes = None
        if ref is not None and bi % 8 == 0:         # the reference's TUs (slower: per-byte push_back)
            ooff, osz, ref_hashes = ref.chunks(blob, "streaming", with_hashes=True)
        else:
            ooff, osz = oracle.chunks(blob, "streaming")
        lo, hi = int(first[bi]), int(first[bi + 1])
        if hi - lo != len(ooff) or not (np.array_equal(co[lo:hi], ooff) and np.array_equal(cs[lo:hi], osz)):
            return f"blob {bi}: boundaries differ"
        mv = memoryview(blob)
        if bd[bi].tobytes() != hashlib.sha256(mv).digest():
            return f"blob {bi}: blob digest differs"
        for j in range(lo, hi):
            o, sz = int(co[j]), int(cs[j])
            if cd[j].tobytes() != hashlib.sha256(mv[o:o + sz]).digest():
                return f"blob {bi}: chunk {j - lo} digest differs"
            if ref_hashes is not None and cd[j].tobytes().hex() != ref_hashes[j - lo]:
                return f"blob {bi}: chunk {j - lo} differs from the reference's Chunk::hash"
        return None

    with ThreadPoolExecutor(max_workers=_oracle.host_threads(128)) as ex:
        bad = [m for m in ex.map(verify, range(n_blobs)) if m]
    assert not bad, bad[:5]
    assert int(first[-1]) == out["n_chunks"] and 300_000 < out["n_chunks"] < 400_000


def _skewed_lengths(rng, total_bytes):
    """The blob-size mix SURVEY.md 8d asks for: 1 KiB ... 64 MiB log-uniform, plus the sizes around the
    chunker's own constants (empty, below the 48-byte window, = min, = max +- 1)."""
    special = [0, 1, 47, 48, 49, 16383, 16384, 16385, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 64 << 20]
    lens, tot = list(special), sum(special)
    while tot < total_bytes:
        n = int(2.0 ** rng.uniform(10, 24.5))
        lens.append(n); tot += n
    order = rng.permutation(len(lens))
    return [lens[i] for i in order]


def _verify_blobs(oracle, blobs, first, co, cs, cd, bd, mode, cfg, every=1):
    import _oracle
    from concurrent.futures import ThreadPoolExecutor

    def verify(bi):
        b = blobs[bi]
        ooff, osz = oracle.chunks(b, mode, **cfg)
        lo, hi = int(first[bi]), int(first[bi + 1])
        if hi - lo != len(ooff) or not (np.array_equal(co[lo:hi], ooff) and np.array_equal(cs[lo:hi], osz)):
            return f"blob {bi} ({len(b)} B): boundaries differ"
        mv = memoryview(b)
        if bd[bi].tobytes() != hashlib.sha256(mv).digest():
            return f"blob {bi} ({len(b)} B): blob digest differs"
        for j in range(lo, hi, every):
            o, n = int(co[j]), int(cs[j])
            if cd[j].tobytes() != hashlib.sha256(mv[o:o + n]).digest():
                return f"blob {bi}: chunk {j - lo} digest differs"
        return None

    with ThreadPoolExecutor(max_workers=_oracle.host_threads(64)) as ex:
        return [m for m in ex.map(verify, range(len(blobs))) if m]


@pytest.mark.parametrize("mode", ["streaming", "rabin"])
def test_ingest_skewed_blob_set_device_and_host_streamed(acc, oracle, mode):
    """A skewed blob set (1 KiB ... 64 MiB log-uniform + the sizes around window / min / max, ~0.6 GiB,
    random bytes at unaligned offsets) through BOTH entry points — device-resident (yams_ingest_device)
    and host-streamed in batches (yams_ingest_host: small batches here, so blobs of every size open and
    close a batch and the 64 MiB blob is a batch of its own) — every blob against the CPU."""
    import torch
    rng = np.random.default_rng(146)
    lens = _skewed_lengths(rng, 600 << 20)
    blobs = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    cfg = {}
    # device-resident: blobs packed back to back (arbitrary byte alignment)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) + 3
    buf = np.concatenate([np.zeros(3, np.uint8)] + blobs + [np.zeros(64, np.uint8)])
    tb = torch.from_numpy(buf).cuda()
    res = acc.ingest_device(tb.data_ptr(), offs, lens, cdc_config(mode), flags=3)
    out = acc.fetch_ingest(res, len(lens))
    bad = _verify_blobs(oracle, blobs, out["blob_first"], out["chunk_offset"], out["chunk_size"], out["chunk_digest"],
                        out["blob_digest"], mode, cfg)
    assert not bad, bad[:5]
    # host-streamed: the same blobs where they lie in (pageable) host memory
    h = acc.ingest_host([b.ctypes.data for b in blobs], lens, cdc_config(mode), flags=3, batch_bytes=24 << 20)
    assert h["n_chunks"] == out["n_chunks"]
    assert np.array_equal(h["blob_first"], out["blob_first"])
    assert np.array_equal(h["chunk_offset"], out["chunk_offset"]) and np.array_equal(h["chunk_size"], out["chunk_size"])
    assert np.array_equal(h["chunk_digest"], out["chunk_digest"]) and np.array_equal(h["blob_digest"], out["blob_digest"])
    # capacity protocol: too few chunk slots -> INVALID_ARG and the required size
    from yams_amd.accel import AccelError
    with pytest.raises(AccelError):
        acc.ingest_host([b.ctypes.data for b in blobs], lens, cdc_config(mode), flags=3, batch_bytes=24 << 20, chunk_cap=10)
    assert acc.last_required_chunks == out["n_chunks"]
    # no blobs at all
    e = acc.ingest_host([], [], cdc_config(mode), flags=3)
    assert e["n_chunks"] == 0 and list(e["blob_first"]) == [0]


def test_cdc_windowed_stream_equals_one_pass(acc, oracle):
    """yams_cdc_chunk_window_host — the device half of StreamingChunker::processStream (streaming_chunker.h:54-121):
    a stream consumed in windows, each window = [64 bytes of history][the open chunk][new bytes], must produce exactly
    the chunks the oracle finds in ONE pass over the whole stream (the reference never resets the rolling hash at a
    chunk boundary).  Windows of odd sizes, histories of exactly 56 / 64 / 200 bytes, four configurations; Rabin mode
    refuses a context."""
    rng = np.random.default_rng(147)
    data = rng.integers(0, 256, 3_000_017, dtype=np.uint8)
    data[1_000_000:1_300_000] = 0                                  # a run without candidates: forced max-size chunks
    for cfg in [dict(min_size=2048, max_size=65536), dict(), dict(min_size=64, max_size=256, mask=0xF),
                dict(min_size=1, max_size=5000, mask=0x3FF, window=16)]:
        want_off, want_sz = oracle.chunks(data, "streaming", **cfg)
        c = cdc_config("streaming", **cfg)
        for window, hist in [(262_147, 64), (1 << 20, 56), (65_536, 200)]:
            got_off, got_sz, got_hx = [], [], []
            start = 0            # stream offset of the open chunk
            end = 0              # stream bytes read so far
            while True:
                end = min(len(data), end + window)
                eof = end == len(data)
                h = min(hist, start)
                off, sz, hx = acc.chunk(data[start - h:end], c, with_hashes=True, context_len=h)
                keep = len(off) if eof else len(off) - 1
                for i in range(keep):
                    got_off.append(start + int(off[i]) - h); got_sz.append(int(sz[i])); got_hx.append(hx[i])
                if eof:
                    break
                start = start + int(off[-1]) - h
            assert got_off == [int(x) for x in want_off] and got_sz == [int(x) for x in want_sz], (cfg, window, hist)
            for i in (0, len(got_off) // 2, len(got_off) - 1):
                assert got_hx[i] == hashlib.sha256(data[got_off[i]:got_off[i] + got_sz[i]].tobytes()).hexdigest()
    with pytest.raises(_lib.AccelError) as e:
        acc.chunk(data[:100000], cdc_config("rabin"), context_len=64)
    assert e.value.status == _lib.YAMS_ERR_INVALID_ARG
    with pytest.raises(_lib.AccelError):
        acc.chunk(data[:100], cdc_config("streaming"), context_len=101)


def test_ingest_defers_the_whole_blob_digest_of_long_blobs(acc, oracle):
    """YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS: blobs above yams_ingest_defer_threshold_{device,host}
    (a pure function of the call's total bytes) come back with an all-zero whole-blob digest — the caller's host hasher
    fills them — while their chunk tables and chunk digests, and everything about the other blobs, are unchanged."""
    import torch
    rng = np.random.default_rng(148)
    lens = [20 << 20, 3 << 20, 1 << 20, (1 << 20) + 1, 700_000, 0, 1, 5000, 2 << 20, 900_000]
    blobs = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    total = sum(lens)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    tb = torch.from_numpy(np.concatenate(blobs + [np.zeros(64, np.uint8)])).cuda()
    cfg = cdc_config("streaming")
    full = acc.fetch_ingest(acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=3), len(lens))
    for thr, run in [(max(1 << 20, total >> 12), lambda: acc.fetch_ingest(acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=7), len(lens))),
                     (max(1 << 20, total >> 9), lambda: acc.ingest_host([b.ctypes.data for b in blobs], lens, cfg, flags=7, batch_bytes=8 << 20))]:
        got = run()
        assert np.array_equal(got["blob_first"], full["blob_first"]) and np.array_equal(got["chunk_offset"], full["chunk_offset"])
        assert np.array_equal(got["chunk_size"], full["chunk_size"]) and np.array_equal(got["chunk_digest"], full["chunk_digest"])
        n_deferred = 0
        for i, n in enumerate(lens):
            if n > thr:
                assert not got["blob_digest"][i].any(), i; n_deferred += 1
            else:
                assert got["blob_digest"][i].tobytes() == hashlib.sha256(blobs[i].tobytes()).digest(), i
        assert n_deferred >= 1
    # without the flag nothing is deferred
    assert all(full["blob_digest"][i].tobytes() == hashlib.sha256(blobs[i].tobytes()).digest() for i in range(len(lens)))


def test_ingest_reference_benchmark_config_4k_16k_64k(acc, oracle):
    """The reference's own chunking benchmark configuration (tests/benchmarks/core_benchmarks.cpp:
    225-229: RabinChunker, min 4096 / target 16384 / max 65536, 1 MiB inputs): 512 such blobs, every one
    against the CPU."""
    import torch
    n_blobs, blen = 512, 1 << 20
    tb = torch.empty(n_blobs * blen, dtype=torch.uint8, device="cuda")
    acc.synth_bytes(7, 0, n_blobs, blen, tb.data_ptr())
    cfg = dict(min_size=4096, max_size=65536)
    res = acc.ingest_device(tb.data_ptr(), [i * blen for i in range(n_blobs)], [blen] * n_blobs, cdc_config("rabin", **cfg), flags=3)
    out = acc.fetch_ingest(res, n_blobs)
    blobs = [oracle.synth_bytes(7, bi, 0, blen) for bi in range(n_blobs)]
    bad = _verify_blobs(oracle, blobs, out["blob_first"], out["chunk_offset"], out["chunk_size"], out["chunk_digest"],
                        out["blob_digest"], "rabin", cfg)
    assert not bad, bad[:5]
    sizes = out["chunk_size"]
    assert sizes.max() <= 65536 and 8000 < sizes.mean() < 20000


def test_chunker_vtable(accel_lib, oracle):
    L = accel_lib
    assert L.yams_plugin_init(b"{}", None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"chunker_v1", 1, C.byref(p)) == 0
    vt = C.cast(p, C.POINTER(_lib.ChunkerV1)).contents
    cfg = _lib.CdcConfig()
    assert vt.get_default_config(None, _lib.CDC_STREAMING, C.byref(cfg)) == 0
    cfg.min_size, cfg.max_size = 2048, 16384
    data = np.random.default_rng(47).integers(0, 256, 500000, dtype=np.uint8)
    chunks = C.POINTER(_lib.ChunkRef)(); n = C.c_size_t()
    assert vt.chunk_data(None, data.ctypes.data_as(_lib.u8p), data.size, C.byref(cfg), C.byref(chunks), C.byref(n)) == 0
    ooff, osz = oracle.chunks(data, "streaming", min_size=2048, max_size=16384)
    assert n.value == len(ooff)
    for i in range(n.value):
        assert (chunks[i].offset, chunks[i].size) == (int(ooff[i]), int(osz[i]))
        assert chunks[i].hash_hex.decode() == hashlib.sha256(data[int(ooff[i]):int(ooff[i] + osz[i])].tobytes()).hexdigest()
    vt.free_chunks(None, chunks, n)
    # chunk_many (version 2): a batch of buffers in ONE device call — the batched ingest path behind the plugin door
    rng = np.random.default_rng(48)
    bufs = [rng.integers(0, 256, n_, dtype=np.uint8) for n_ in (300000, 0, 47, 2048, 2049, 1 << 20, 1, 70001)]
    bufs[3][:] = 0                                        # a constant buffer: no candidates, forced cuts only
    ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * len(bufs))(*[b.size for b in bufs])
    for flags in (0, _lib.CHUNK_MANY_BUFFER_HASHES):
        batch = C.POINTER(_lib.ChunkBatch)()
        assert vt.chunk_many(None, ptrs, lens, len(bufs), C.byref(cfg), flags, C.byref(batch)) == 0
        bt = batch.contents
        assert bt.n_buffers == len(bufs) and bt.first_chunk[0] == 0 and bt.first_chunk[len(bufs)] == bt.n_chunks
        assert bool(bt.buffer_hash_hex) == bool(flags)
        for b, data_b in enumerate(bufs):
            ooff, osz = oracle.chunks(data_b, "streaming", min_size=2048, max_size=16384)
            lo, hi = bt.first_chunk[b], bt.first_chunk[b + 1]
            assert hi - lo == len(ooff), (b, hi - lo, len(ooff))
            for i in range(len(ooff)):
                ch = bt.chunks[lo + i]
                assert (ch.offset, ch.size) == (int(ooff[i]), int(osz[i]))
                if i % 7 == 0 or i == len(ooff) - 1:
                    assert ch.hash_hex.decode() == hashlib.sha256(data_b[int(ooff[i]):int(ooff[i] + osz[i])].tobytes()).hexdigest()
            if flags:
                assert C.string_at(C.addressof(bt.buffer_hash_hex.contents) + 65 * b).decode() == hashlib.sha256(data_b.tobytes()).hexdigest()
        vt.free_chunk_batch(None, batch)
    batch = C.POINTER(_lib.ChunkBatch)()
    assert vt.chunk_many(None, None, None, 0, C.byref(cfg), 1, C.byref(batch)) == 0 and batch.contents.n_chunks == 0
    vt.free_chunk_batch(None, batch)
    assert vt.chunk_many(None, ptrs, lens, len(bufs), None, 0, C.byref(batch)) == _lib.YAMS_ERR_INVALID_ARG
    L.yams_plugin_shutdown()


def test_content_hash_vtable_refuses_lone_long_chains(accel_lib):
    """One SHA-256 chain runs at ~35 MB/s on a device lane and > 1 GB/s on a host core: hash() above
    YAMS_HASH_LONE_CHAIN_MAX and batches dominated by one long message return YAMS_ERR_UNSUPPORTED (the host hashes
    those itself, abi_model_provider_adapter.cpp:121-122); what the device is good at is served."""
    L = accel_lib
    assert L.yams_plugin_init(b"{}", None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"content_hash_v1", 1, C.byref(p)) == 0
    vt = C.cast(p, C.POINTER(_lib.ContentHashV1)).contents
    rng = np.random.default_rng(49)
    big = rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    out = C.create_string_buffer(65)
    assert vt.hash(None, big.ctypes.data_as(_lib.u8p), big.size, out) == _lib.YAMS_ERR_UNSUPPORTED
    assert vt.hash(None, big.ctypes.data_as(_lib.u8p), _lib.HASH_LONE_CHAIN_MAX, out) == 0
    assert out.value.decode() == hashlib.sha256(big[:_lib.HASH_LONE_CHAIN_MAX].tobytes()).hexdigest()
    # 1 x 3 MiB + 8 x 4 KiB: refused; 150 x 1 MiB + 1 x 2.5 MiB: served (2.5 MiB < 152.5 MiB / 37)
    def many(sizes):
        ptrs = (_lib.u8p * len(sizes))(*[C.cast(big.ctypes.data + (i * 4099) % 1000, _lib.u8p) for i in range(len(sizes))])
        lens = (C.c_size_t * len(sizes))(*sizes)
        hexes = C.create_string_buffer(65 * len(sizes))
        st = vt.hash_many(None, ptrs, lens, len(sizes), hexes)
        return st, [hexes.raw[65 * i:65 * i + 64].decode() for i in range(len(sizes))]
    assert many([(3 << 20) - 1000] + [4096] * 8)[0] == _lib.YAMS_ERR_UNSUPPORTED
    st, hx = many([1 << 20] * 150 + [5 << 19])
    assert st == 0
    for i in (0, 77, 150):
        o = (i * 4099) % 1000
        n_ = (1 << 20) if i < 150 else (5 << 19)
        assert hx[i] == hashlib.sha256(big[o:o + n_].tobytes()).hexdigest()
    hp = C.c_void_p()
    assert L.yams_plugin_get_health_json(C.byref(hp)) == 0
    import json
    assert json.loads(C.string_at(hp))["refused_lone_chains"] >= 2
    C.CDLL(None).free(hp)
    L.yams_plugin_shutdown()


def test_randomised_ingest_stress_against_cpu_checkers():
    """tests/stress_ingest.py: random ragged blob sets (random / constant / periodic / text-like
    content at arbitrary byte offsets), random chunker configurations, both chunkers; boundaries,
    chunk digests and blob digests bit-exact against the oracle + hashlib.  (220 further cases,
    4.9 M chunks, were run by hand with seeds 1-4: no mismatch.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_ingest.py"), "--cases", "30", "--seed", "9"],
                       capture_output=True, text=True, timeout=280)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1])
    assert res["cases"] == 30 and res["mismatches"] == 0 and res["chunks"] > 1000, res
