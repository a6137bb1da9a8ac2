"""ctypes doors into the CHECKERS under oracle/ — test infrastructure only.

`oracle` = the plain-C restatement (oracle/yams_oracle.c, always buildable);
`ref`    = the reference's own translation units (oracle/_ref/libyams_ref.so, built in the dev
           container from /root/reference; travels prebuilt to the GPU box).  May be None.
Nothing under yams_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libyams_oracle.so")
_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libyams_ref.so")
_SCAN_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libyams_scan_ref.so")

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)


class CdcConfig(C.Structure):
    _fields_ = [("window_size", C.c_uint64), ("min_size", C.c_uint64), ("max_size", C.c_uint64),
                ("polynomial", C.c_uint64), ("mask", C.c_uint64)]


DEFAULT_CDC = dict(window=48, min_size=16 * 1024, max_size=1024 * 1024,
                   polynomial=0x3DA3358B4DC173, mask=0x1FFF)


def build():
    src = os.path.join(ORACLE_DIR, "yams_oracle.c")
    stale = (not os.path.exists(_ORACLE_SO)) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(src)
    need_ref = os.path.isdir("/root/reference/src") and not (os.path.exists(_REF_SO) and os.path.exists(_SCAN_REF_SO))
    if not need_ref and os.path.isdir("/root/reference/src"):
        wrap = os.path.join(ORACLE_DIR, "scan_ref_wrap.cpp")
        need_ref = os.path.getmtime(_SCAN_REF_SO) < max(os.path.getmtime(wrap), os.path.getmtime(os.path.join(ORACLE_DIR, "gen_scan_ref.py")))
    if stale or need_ref:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self):
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        L.oracle_sha256.argtypes = [u8p, C.c_size_t, u8p]
        L.oracle_sha256_hex.argtypes = [u8p, C.c_size_t, C.c_char_p]
        L.oracle_rabin_table.argtypes = [C.c_uint64, u64p]
        for f in (L.oracle_chunk_rabin, L.oracle_chunk_streaming):
            f.argtypes = [u8p, C.c_size_t, C.POINTER(CdcConfig), u64p, u64p, C.c_size_t]
            f.restype = C.c_size_t
        L.oracle_query_invalid.argtypes = [f32p, C.c_size_t]
        L.oracle_query_invalid.restype = C.c_int
        L.oracle_exact_scan_cosine.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t,
                                               C.c_float, u64p, i64p, f32p, u64p, u64p]
        L.oracle_exact_scan_cosine.restype = C.c_long
        L.oracle_exact_scan_cosine_records.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t, C.c_int,
                                                       C.c_float, u64p, C.c_void_p, i64p, f32p, u64p]
        L.oracle_exact_scan_cosine_records.restype = C.c_long
        L.oracle_exact_scan_l2.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t,
                                           C.c_float, u64p, i64p, f32p, f32p]
        L.oracle_exact_scan_l2.restype = C.c_long
        u32p_ = C.POINTER(C.c_uint32)
        L.oracle_exact_scan_cosine_many.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t, C.c_size_t, C.c_float,
                                                    i64p, f32p, u32p_]
        L.oracle_exact_scan_cosine_many.restype = C.c_long
        L.oracle_exact_scan_l2_many.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t, C.c_size_t, i64p, f32p, f32p, u32p_]
        L.oracle_exact_scan_l2_many.restype = C.c_long
        L.oracle_set_lanes.argtypes = [C.c_int]
        L.oracle_set_lanes.restype = C.c_int
        L.oracle_exact_scan_l2_f32acc.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t, C.c_float, u64p, C.c_int,
                                                  i64p, f32p, f32p]
        L.oracle_exact_scan_l2_f32acc.restype = C.c_long
        L.oracle_l2_distance_f32acc_many.argtypes = [f32p, f32p, C.c_size_t, C.c_size_t, C.c_int, f32p]
        L.oracle_l2_distance_f32acc_many.restype = None
        L.oracle_cosine_similarity.argtypes = [f32p, f32p, C.c_size_t]
        L.oracle_cosine_similarity.restype = C.c_double
        L.oracle_philox4x32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
        L.oracle_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, f32p]
        L.oracle_synth_bytes.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, u8p]
        L.oracle_mt19937_rows.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, f32p]

    # --- SHA-256 ---
    def sha256_hex(self, data: bytes | np.ndarray) -> str:
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        a = np.ascontiguousarray(a, dtype=np.uint8)
        out = C.create_string_buffer(65)
        self.L.oracle_sha256_hex(_ptr(a, u8p) if a.size else None, a.size, out)
        return out.value.decode()

    # --- CDC ---
    def chunks(self, data, mode="streaming", **cfg):
        c = dict(DEFAULT_CDC); c.update(cfg)
        a = np.ascontiguousarray(data if isinstance(data, np.ndarray)
                                 else np.frombuffer(bytes(data), dtype=np.uint8), dtype=np.uint8)
        conf = CdcConfig(c["window"], c["min_size"], c["max_size"], c["polynomial"], c["mask"])
        floor = max(1, min(c["min_size"], c["max_size"]))
        cap = a.size // floor + 2
        off = np.zeros(cap, np.uint64); sz = np.zeros(cap, np.uint64)
        fn = self.L.oracle_chunk_streaming if mode == "streaming" else self.L.oracle_chunk_rabin
        n = fn(_ptr(a, u8p) if a.size else None, a.size, C.byref(conf), _ptr(off, u64p),
               _ptr(sz, u64p), cap)
        assert n <= cap
        return off[:n].copy(), sz[:n].copy()

    def rabin_table(self, poly):
        t = np.zeros(256, np.uint64)
        self.L.oracle_rabin_table(poly, _ptr(t, u64p))
        return t

    # --- exact scan ---
    def scan_cosine(self, corpus, query, k, thr=-1.0, tie_rank=None):
        corpus = np.ascontiguousarray(corpus, np.float32); query = np.ascontiguousarray(query, np.float32)
        n, d = corpus.shape
        rows = np.full(max(k, 1), -1, np.int64); sims = np.zeros(max(k, 1), np.float32)
        rv = C.c_uint64(0); ev = C.c_uint64(0)
        tr = None
        if tie_rank is not None:
            tie_rank = np.ascontiguousarray(tie_rank, np.uint64); tr = _ptr(tie_rank, u64p)
        cnt = self.L.oracle_exact_scan_cosine(_ptr(corpus, f32p), n, d, _ptr(query, f32p), k, thr,
                                              tr, _ptr(rows, i64p), _ptr(sims, f32p),
                                              C.byref(rv), C.byref(ev))
        if cnt < 0:
            return None
        return rows[:cnt].copy(), sims[:cnt].copy(), rv.value, ev.value

    def scan_cosine_many(self, corpus, queries, k, thr=-1.0):
        """Every query of a batch against one slice (oracle_exact_scan_cosine_many: the single-query arithmetic, the
        queries in the lanes of a vector).  Returns (rows [nq][k] int64, sims [nq][k] float32, counts [nq]) with rows
        -1 past a query's count, or None when a query is invalid."""
        corpus = np.ascontiguousarray(corpus, np.float32); queries = np.atleast_2d(np.ascontiguousarray(queries, np.float32))
        n, d = corpus.shape
        nq = queries.shape[0]
        rows = np.full((nq, max(k, 1)), -1, np.int64); sims = np.zeros((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        rc = self.L.oracle_exact_scan_cosine_many(_ptr(corpus, f32p), n, d, _ptr(queries, f32p), nq, k, thr, _ptr(rows, i64p),
                                                  _ptr(sims, f32p), counts.ctypes.data_as(C.POINTER(C.c_uint32)))
        return None if rc < 0 else (rows, sims, counts)

    def set_lanes(self, bits=0):
        """Pin the vector width of the batched drivers (128 / 256 / 512 bits; 0 = the widest the host has); returns the
        width in effect."""
        return int(self.L.oracle_set_lanes(bits))

    def scan_l2_many(self, corpus, queries, k):
        """The k nearest rows of every query (distance asc, row asc) and their cosine — the vec0 cut; the cosine threshold
        is the caller's.  Returns (rows, dist, sims, counts)."""
        corpus = np.ascontiguousarray(corpus, np.float32); queries = np.atleast_2d(np.ascontiguousarray(queries, np.float32))
        n, d = corpus.shape
        nq = queries.shape[0]
        rows = np.full((nq, max(k, 1)), -1, np.int64); dist = np.zeros((nq, max(k, 1)), np.float32)
        sims = np.zeros((nq, max(k, 1)), np.float32); counts = np.zeros(nq, np.uint32)
        self.L.oracle_exact_scan_l2_many(_ptr(corpus, f32p), n, d, _ptr(queries, f32p), nq, k, _ptr(rows, i64p), _ptr(dist, f32p),
                                         _ptr(sims, f32p), counts.ctypes.data_as(C.POINTER(C.c_uint32)))
        return rows, dist, sims, counts

    def scan_cosine_records(self, corpus, query, k, thr=-1.0, tie_rank=None, allow=None, all_matching=False):
        """The reference's record path (metadata_filters), sqlite_vec_backend.cpp:4333-4409."""
        corpus = np.ascontiguousarray(corpus, np.float32); query = np.ascontiguousarray(query, np.float32)
        n, d = corpus.shape
        cap = max(n if all_matching else k, 1)
        rows = np.full(cap, -1, np.int64); sims = np.zeros(cap, np.float32)
        ev = C.c_uint64(0)
        tr = al = None
        if tie_rank is not None:
            tie_rank = np.ascontiguousarray(tie_rank, np.uint64); tr = _ptr(tie_rank, u64p)
        if allow is not None:
            allow = np.ascontiguousarray(allow, np.uint8); al = allow.ctypes.data_as(C.c_void_p)
        cnt = self.L.oracle_exact_scan_cosine_records(_ptr(corpus, f32p), n, d, _ptr(query, f32p), k,
                                                      1 if all_matching else 0, thr, tr, al,
                                                      _ptr(rows, i64p), _ptr(sims, f32p), C.byref(ev))
        if cnt < 0:
            return None
        return rows[:cnt].copy(), sims[:cnt].copy(), ev.value

    def scan_l2(self, corpus, query, k, thr=-1.0, tie_rank=None):
        corpus = np.ascontiguousarray(corpus, np.float32); query = np.ascontiguousarray(query, np.float32)
        n, d = corpus.shape
        rows = np.full(max(k, 1), -1, np.int64); dist = np.zeros(max(k, 1), np.float32)
        sims = np.zeros(max(k, 1), np.float32)
        tr = None
        if tie_rank is not None:
            tie_rank = np.ascontiguousarray(tie_rank, np.uint64); tr = _ptr(tie_rank, u64p)
        cnt = self.L.oracle_exact_scan_l2(_ptr(corpus, f32p), n, d, _ptr(query, f32p), k, thr, tr,
                                          _ptr(rows, i64p), _ptr(dist, f32p), _ptr(sims, f32p))
        return rows[:cnt].copy(), dist[:cnt].copy(), sims[:cnt].copy()

    def scan_l2_f32acc(self, corpus, query, k, thr=-1.0, tie_rank=None, lanes=1):
        """The vec0 scan under the OTHER plausible definition: fp32 accumulation (sequential, or `lanes` SIMD-style
        partial sums) — what the absent sqlite-vec-cpp most likely does.  For reports, not for parity."""
        corpus = np.ascontiguousarray(corpus, np.float32); query = np.ascontiguousarray(query, np.float32)
        n, d = corpus.shape
        rows = np.full(max(k, 1), -1, np.int64); dist = np.zeros(max(k, 1), np.float32)
        sims = np.zeros(max(k, 1), np.float32)
        tr = None
        if tie_rank is not None:
            tie_rank = np.ascontiguousarray(tie_rank, np.uint64); tr = _ptr(tie_rank, u64p)
        cnt = self.L.oracle_exact_scan_l2_f32acc(_ptr(corpus, f32p), n, d, _ptr(query, f32p), k, thr, tr, lanes,
                                                 _ptr(rows, i64p), _ptr(dist, f32p), _ptr(sims, f32p))
        return rows[:cnt].copy(), dist[:cnt].copy(), sims[:cnt].copy()

    def l2_f32acc_many(self, rows, query, lanes=1):
        rows = np.ascontiguousarray(rows, np.float32); query = np.ascontiguousarray(query, np.float32)
        out = np.empty(rows.shape[0], np.float32)
        self.L.oracle_l2_distance_f32acc_many(_ptr(rows, f32p), _ptr(query, f32p), rows.shape[0], rows.shape[1], lanes, _ptr(out, f32p))
        return out

    def pq_search(self, corpus, codes, lut, query, k, thr=-1.0, rerank_factor=2, tie_keys=None, row_of_index=None, chunk_rank=None,
                  candidates=None, sum_lanes=1):
        """simeonPqSearchUnlocked restated (oracle_pq_search; PARITY UNPINNED for the order of the ADC sum): returns
        (rows, sims, {"candidates", "materialised"})."""
        corpus = np.ascontiguousarray(corpus, np.float32); codes = np.ascontiguousarray(codes, np.uint8)
        lut = np.ascontiguousarray(lut, np.float32); query = np.ascontiguousarray(query, np.float32)
        n_rows, dim = corpus.shape
        n, m = codes.shape
        rows = np.full(max(k, 1), -1, np.int64); sims = np.zeros(max(k, 1), np.float32)
        u8p = C.POINTER(C.c_uint8); u32p = C.POINTER(C.c_uint32)
        tk = roi = cr = cd = None
        if tie_keys is not None:
            tie_keys = np.ascontiguousarray(tie_keys, np.uint64); tk = _ptr(tie_keys, u64p)
        if row_of_index is not None:
            row_of_index = np.ascontiguousarray(row_of_index, np.uint32); roi = _ptr(row_of_index, u32p)
        if chunk_rank is not None:
            chunk_rank = np.ascontiguousarray(chunk_rank, np.uint64); cr = _ptr(chunk_rank, u64p)
        n_c = 0
        if candidates is not None:
            candidates = np.ascontiguousarray(candidates, np.uint32); n_c = candidates.size
            cd = _ptr(candidates if n_c else np.zeros(1, np.uint32), u32p)
        st = (C.c_uint64 * 2)()
        f = self.L.oracle_pq_search
        f.restype = C.c_long
        f.argtypes = [f32p, C.c_size_t, C.c_size_t, u8p, C.c_size_t, C.c_size_t, f32p, u64p, u32p, u64p, f32p, C.c_size_t, C.c_float,
                      C.c_size_t, u32p, C.c_size_t, C.c_int, i64p, f32p, C.POINTER(C.c_uint64)]
        cnt = f(_ptr(corpus, f32p), n_rows, dim, _ptr(codes, u8p), n, m, _ptr(lut, f32p), tk, roi, cr, _ptr(query, f32p), k, thr,
                rerank_factor, cd, n_c, sum_lanes, _ptr(rows, i64p), _ptr(sims, f32p), st)
        return rows[:cnt].copy(), sims[:cnt].copy(), {"candidates": int(st[0]), "materialised": int(st[1])}

    def pq_adc_score(self, code, lut, sum_lanes=1):
        code = np.ascontiguousarray(code, np.uint8); lut = np.ascontiguousarray(lut, np.float32)
        f = self.L.oracle_pq_adc_score
        f.restype = C.c_float
        f.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, f32p, C.c_int]
        return float(f(_ptr(code, C.POINTER(C.c_uint8)), code.size, _ptr(lut, f32p), sum_lanes))

    def cosine(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        return self.L.oracle_cosine_similarity(_ptr(a, f32p), _ptr(b, f32p), a.size)

    # --- synthetic data ---
    def synth_rows(self, seed, row0, n, dim):
        out = np.empty((n, dim), np.float32)
        self.L.oracle_synth_rows(seed, row0, n, dim, _ptr(out, f32p))
        return out

    def mt19937_rows(self, seed, skip_vectors, n, dim):
        """The reference's own recipe (vector_backend_engine_compare.cpp:83-107): std::mt19937(seed),
        U(-1,1) floats, fp32 normalise; `skip_vectors` vectors of the stream are skipped first."""
        out = np.empty((n, dim), np.float32)
        self.L.oracle_mt19937_rows(seed, skip_vectors, n, dim, _ptr(out, f32p))
        return out

    def synth_bytes(self, seed, blob, off, n):
        out = np.empty(n, np.uint8)
        if n:
            self.L.oracle_synth_bytes(seed, blob, off, n, _ptr(out, u8p))
        return out

    def philox(self, seed, lo, hi):
        o = (C.c_uint32 * 4)()
        self.L.oracle_philox4x32(seed, lo, hi, o)
        return list(o)


class Ref:
    """The reference's own TUs.  Raises FileNotFoundError when the prebuilt .so is absent."""

    def __init__(self):
        build()
        if not os.path.exists(_REF_SO):
            raise FileNotFoundError(_REF_SO)
        L = C.CDLL(_REF_SO)
        self.L = L
        L.ref_sha256_hex.argtypes = [u8p, C.c_size_t, C.c_char_p]
        L.ref_sha256_hex_split.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t,
                                           C.c_char_p]
        for f in (L.ref_chunk_rabin, L.ref_chunk_streaming):
            f.argtypes = [u8p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                          C.c_uint64, u64p, u64p, C.c_char_p, C.c_size_t]
            f.restype = C.c_size_t

    def mt19937_rows(self, seed, n, dim):
        out = np.empty((n, dim), np.float32)
        self.L.ref_mt19937_rows.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t, f32p]
        self.L.ref_mt19937_rows(seed, n, dim, _ptr(out, f32p))
        return out

    def sha256_hex(self, data) -> str:
        a = np.ascontiguousarray(data if isinstance(data, np.ndarray)
                                 else np.frombuffer(bytes(data), dtype=np.uint8), dtype=np.uint8)
        out = C.create_string_buffer(65)
        self.L.ref_sha256_hex(_ptr(a, u8p) if a.size else None, a.size, out)
        return out.value.decode()

    def sha256_hex_split(self, data, cuts) -> str:
        a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        arr = (C.c_size_t * len(cuts))(*cuts)
        out = C.create_string_buffer(65)
        self.L.ref_sha256_hex_split(_ptr(a, u8p) if a.size else None, a.size, arr, len(cuts), out)
        return out.value.decode()

    def chunks(self, data, mode="streaming", with_hashes=True, **cfg):
        c = dict(DEFAULT_CDC); c.update(cfg)
        a = np.ascontiguousarray(data if isinstance(data, np.ndarray)
                                 else np.frombuffer(bytes(data), dtype=np.uint8), dtype=np.uint8)
        floor = max(1, min(c["min_size"], c["max_size"]))
        cap = a.size // floor + 2
        off = np.zeros(cap, np.uint64); sz = np.zeros(cap, np.uint64)
        hexbuf = C.create_string_buffer(65 * cap) if with_hashes else None
        fn = self.L.ref_chunk_streaming if mode == "streaming" else self.L.ref_chunk_rabin
        n = fn(_ptr(a, u8p) if a.size else None, a.size, c["window"], c["min_size"], c["max_size"],
               c["polynomial"], c["mask"], _ptr(off, u64p), _ptr(sz, u64p), hexbuf, cap)
        assert n <= cap
        hashes = None
        if with_hashes:
            raw = hexbuf.raw
            hashes = [raw[65 * i:65 * i + 64].decode() for i in range(n)]
        return off[:n].copy(), sz[:n].copy(), hashes


class ScanRef:
    """The reference's OWN exact-scan loop (oracle/_ref/libyams_scan_ref.so: bruteForceSearchUnlocked + helpers cut
    verbatim from /root/reference by oracle/gen_scan_ref.py, over an in-memory SQLite).  One instance = one `vectors`
    table.  Raises FileNotFoundError when the prebuilt .so is absent."""

    def __init__(self):
        build()
        if not os.path.exists(_SCAN_REF_SO):
            raise FileNotFoundError(_SCAN_REF_SO)
        try:    # built with the reference's '-mavx', '-mfma' (oracle/Makefile): only for hosts that have them
            flags = open("/proc/cpuinfo").read()
            if " avx" not in flags or " fma" not in flags:
                raise FileNotFoundError("this host lacks AVX / FMA: " + _SCAN_REF_SO + " cannot run here")
        except OSError:
            pass
        L = C.CDLL(_SCAN_REF_SO)
        self.L = L
        L.scanref_open.restype = C.c_void_p
        L.scanref_close.argtypes = [C.c_void_p]
        L.scanref_insert.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_char_p]
        L.scanref_insert_rows.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_longlong]
        L.scanref_search.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p, C.c_size_t, C.c_int,
                                     C.POINTER(C.c_longlong), f32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_ulonglong)]
        L.scanref_search.restype = C.c_long
        L.scanref_search_ex.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_char_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_longlong), f32p, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_ulonglong)]
        L.scanref_search_ex.restype = C.c_long
        L.scanref_insert_rows_docs.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_longlong]
        L.scanref_cosine.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t]
        L.scanref_cosine.restype = C.c_double
        # round 6: the in-tree half of the L2 path (vec0SearchUnlocked over the harness's vec0 module)
        self.has_vec0 = hasattr(L, "scanref_vec0_search")
        if self.has_vec0:
            L.scanref_vec0_set_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            L.scanref_vec0_rebuild.argtypes = [C.c_void_p, C.c_size_t]
            L.scanref_vec0_rebuild.restype = C.c_long
            L.scanref_delete_ordinal.argtypes = [C.c_void_p, C.c_longlong]
            L.scanref_rowid_of_ordinal.argtypes = [C.c_void_p, C.c_longlong]
            L.scanref_rowid_of_ordinal.restype = C.c_longlong
            L.scanref_vec0_search.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_float, C.POINTER(C.c_longlong), C.c_size_t,
                                              C.POINTER(C.c_longlong), f32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_ulonglong)]
            L.scanref_vec0_search.restype = C.c_long
        self.invalid_argument = -int(L.scanref_error_code_invalid_argument())
        self.h = C.c_void_p(L.scanref_open())
        if not self.h:
            raise OSError("scanref_open failed")
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.scanref_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def insert_rows(self, rows, chunk_ids=None, document_hashes=None):
        """Rows as the backend stores them (raw fp32 blobs); chunk ids default to zero-padded ordinals (chunk-id order ==
        row order); document_hashes: one per row (default "doc")."""
        rows = np.ascontiguousarray(rows, np.float32)
        n, d = rows.shape
        ids = docs = None
        if chunk_ids is not None:
            ids = (C.c_char_p * n)(*[s.encode() for s in chunk_ids])
        if document_hashes is not None:
            docs = (C.c_char_p * n)(*[s.encode() for s in document_hashes])
        assert self.L.scanref_insert_rows_docs(self.h, _ptr(rows, f32p), n, d, ids, docs, self.n) == 0
        self.n += n

    def insert_raw(self, chunk_id, blob: bytes | None, embedding_dim, metadata=None, document_hash="doc"):
        """One row with an arbitrary blob (wrong sizes, empty) and an optional flat metadata dict."""
        meta = None
        if metadata is not None:
            meta = ("{" + ",".join('"%s":"%s"' % (k, v) for k, v in metadata.items()) + "}").encode()
        assert self.L.scanref_insert(self.h, chunk_id.encode(), document_hash.encode(), blob, len(blob) if blob else 0,
                                     embedding_dim, self.n, meta) == 0
        self.n += 1

    def search(self, query, k, thr=-1.0, metadata_filters=None, all_matching=False, document_hash=None, candidate_hashes=None):
        """bruteForceSearchUnlocked(query, k, thr, document_hash, candidate_hashes, metadata_filters, &diag, TopK |
        AllMatching): returns (ordinals, scores, diag dict) or the negative ErrorCode."""
        q = np.ascontiguousarray(query, np.float32)
        cap = max(self.n if (all_matching or metadata_filters) else k, 1)
        cap = max(cap, k, 1)
        ords = np.full(cap, -1, np.int64); sc = np.zeros(cap, np.float32)
        cnt = C.c_size_t(0); dg = (C.c_ulonglong * 4)()
        kv = None; n_meta = 0
        if metadata_filters:
            flat = []
            for kk, vv in metadata_filters.items():
                flat += [kk.encode(), vv.encode()]
            kv = (C.c_char_p * len(flat))(*flat); n_meta = len(metadata_filters)
        cand = None; n_cand = 0
        if candidate_hashes:
            cl = sorted(candidate_hashes)
            cand = (C.c_char_p * len(cl))(*[c.encode() for c in cl]); n_cand = len(cl)
            cap = max(cap, self.n)
            ords = np.full(cap, -1, np.int64); sc = np.zeros(cap, np.float32)
        rc = self.L.scanref_search_ex(self.h, _ptr(q, f32p), q.size, k, thr, document_hash.encode() if document_hash else None,
                                      cand, n_cand, kv, n_meta, 1 if all_matching else 0,
                                      ords.ctypes.data_as(C.POINTER(C.c_longlong)), _ptr(sc, f32p), cap, C.byref(cnt), dg)
        if rc != 0:
            return rc
        assert cnt.value <= cap
        return ords[:cnt.value].copy(), sc[:cnt.value].copy(), {"rows_visited": dg[0], "exact_distance_evaluations": dg[1],
                                                               "returned_rows": dg[2], "used_exact_scan": dg[3]}

    def cosine(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        return self.L.scanref_cosine(_ptr(a, f32p), a.size, _ptr(b, f32p), b.size)

    # ---- the L2 (vec0) path: vec0SearchUnlocked, sqlite_vec_backend.cpp:4450-4530 ------------------------------------------
    def vec0_set_distance(self, lanes=None):
        """The distance the harness's vec0 module computes: None = fp64 (this repository's default definition), else the
        oracle's oracle_l2_distance_f32acc with that `lanes` (1 / 8 / 16 sequential or SIMD-style fp32 sums, negative = fused)."""
        if lanes is None:
            self.L.scanref_vec0_set_distance(self.h, None, 0)
        else:
            fn = C.cast(oracle().L.oracle_l2_distance_f32acc, C.c_void_p)
            self.L.scanref_vec0_set_distance(self.h, fn, int(lanes))

    def vec0_rebuild(self, dim):
        """rebuildVec0DimUnlocked(dim): the reference's own creation + population of the vec0 table from `vectors`."""
        rc = self.L.scanref_vec0_rebuild(self.h, dim)
        assert rc == 0, rc

    def delete_ordinal(self, ordinal):
        assert self.L.scanref_delete_ordinal(self.h, ordinal) == 0

    def rowid_of(self, ordinal):
        return int(self.L.scanref_rowid_of_ordinal(self.h, ordinal))

    def vec0_search(self, query, k, thr=-1.0, candidate_rowids=None):
        """vec0SearchUnlocked(query, k, thr, candidateRowids): (ordinals, cosine scores, stats) or the negative ErrorCode."""
        q = np.ascontiguousarray(query, np.float32)
        cap = max(k, 1)
        ords = np.full(cap, -1, np.int64); sc = np.zeros(cap, np.float32)
        cnt = C.c_size_t(0); st = (C.c_ulonglong * 2)()
        cand = None; n_cand = 0
        if candidate_rowids is not None:
            n_cand = len(candidate_rowids)
            cand = (C.c_longlong * max(n_cand, 1))(*[int(x) for x in candidate_rowids])
        rc = self.L.scanref_vec0_search(self.h, _ptr(q, f32p), q.size, k, thr, cand, n_cand,
                                        ords.ctypes.data_as(C.POINTER(C.c_longlong)), _ptr(sc, f32p), cap, C.byref(cnt), st)
        if rc != 0:
            return rc
        assert cnt.value <= cap
        return ords[:cnt.value].copy(), sc[:cnt.value].copy(), {"knn_queries": st[0], "rowid_probes": st[1]}


def scan_ref():
    """A fresh ScanRef (its own in-memory table), or None when oracle/_ref/libyams_scan_ref.so did not travel."""
    try:
        return ScanRef()
    except (FileNotFoundError, OSError):
        return None


_oracle = None
_ref = None


def oracle() -> Oracle:
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def ref():
    global _ref
    if _ref is None:
        try:
            _ref = Ref()
        except (FileNotFoundError, OSError):
            _ref = False
    return _ref or None


# ---- the oracle over all host cores ------------------------------------------------------------------
def host_threads(limit: int | None = None) -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a
    container that shows 256 CPUs may be allowed 16 CPU-seconds per second)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, min(n, limit) if limit else n)


MANY_FROM = 8   # scan_threaded: batches of at least this many queries go through the batched drivers


def scan_threaded(get_slice, n_rows, queries, k, metric="cosine", thr=-1.0, slice_rows=65536,
                  threads=None, stats=None):
    """The scalar oracle scan of every query over a corpus that need not fit in host memory:
    `get_slice(lo, hi)` returns rows [lo, hi) as a float32 array (a device-to-host copy of the very
    tensor the GPU scanned, or a regenerated Philox slice).  One task = one row slice x all queries
    (ctypes calls release the GIL, so the tasks run on all host cores); per-slice top-k lists are
    merged with the reference comparator (similarity desc / distance asc, then row id asc) — the
    exact top-k of a union is the top-k of the per-part top-k lists.  Returns one
    (rows, sims[, dist]) tuple per query, identical to a single oracle call over the whole corpus."""
    from concurrent.futures import ThreadPoolExecutor
    import time
    o = oracle()
    queries = np.atleast_2d(np.ascontiguousarray(queries, np.float32))
    nq = queries.shape[0]
    threads = threads or host_threads()
    bounds = [(lo, min(lo + slice_rows, n_rows)) for lo in range(0, n_rows, slice_rows)]

    def task(b):
        lo, hi = b
        t0 = time.perf_counter()
        part = np.ascontiguousarray(get_slice(lo, hi), np.float32)
        t1 = time.perf_counter()
        out = []
        many = None
        if nq >= MANY_FROM:     # the batched drivers (pinned against the single-query functions by test_oracle.py)
            many = o.scan_cosine_many(part, queries, k, thr) if metric == "cosine" else o.scan_l2_many(part, queries, k)
        if many is not None and metric == "cosine":
            rows, sims, counts = many
            out = [(rows[qi, :counts[qi]] + lo, sims[qi, :counts[qi]], None) for qi in range(nq)]
        elif many is not None:
            rows, dist, sims, counts = many
            out = [(rows[qi, :counts[qi]] + lo, sims[qi, :counts[qi]], dist[qi, :counts[qi]]) for qi in range(nq)]
        else:
            for qi in range(nq):
                if metric == "cosine":
                    rows, sims, _, _ = o.scan_cosine(part, queries[qi], k, thr)
                    out.append((rows + lo, sims, None))
                else:
                    rows, dist, sims = o.scan_l2(part, queries[qi], k, -1.0)   # threshold after the merge
                    out.append((rows + lo, sims, dist))
        return out, t1 - t0, time.perf_counter() - t1

    t_start = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(task, bounds))
    wall = time.perf_counter() - t_start
    if stats is not None:
        stats.update({"threads": threads, "slices": len(bounds), "wall_s": wall,
                      "fetch_thread_s": sum(p[1] for p in parts), "scan_thread_s": sum(p[2] for p in parts)})
    res = []
    for qi in range(nq):
        rows = np.concatenate([p[0][qi][0] for p in parts])
        sims = np.concatenate([p[0][qi][1] for p in parts])
        if metric == "cosine":
            order = np.lexsort((rows, -sims.astype(np.float64)))[:k]
            res.append((rows[order], sims[order]))
        else:
            dist = np.concatenate([p[0][qi][2] for p in parts])
            order = np.lexsort((rows, dist.astype(np.float64)))[:k]
            rows, sims, dist = rows[order], sims[order], dist[order]
            keep = ~(sims < np.float32(thr))      # vec0: k nearest, THEN the cosine threshold (:4506-4510)
            res.append((rows[keep], sims[keep], dist[keep]))
    return res


def l2_definition_report(o, fetch_rows, queries, rows64, dist64, k):
    """How much the choice of the L2 accumulation precision matters.  `rows64` / `dist64` [nq][K]: the top
    K > k rows of every query under THIS repository's definition (fp64 accumulate), e.g. from the device; for those
    candidates the distance is recomputed with fp32 accumulation (sequential, 8 and 16 SIMD-style lanes) and the top-k
    SET and ORDER under each variant are compared with the fp64 ones.  Conclusive for a query when no candidate beyond
    rank K could overtake the cut: the fp64 gap between rank k and rank K exceeds twice the largest measured
    |d32 - d64| of the query."""
    nq, K = rows64.shape
    out = {"queries": int(nq), "k": int(k), "candidates_per_query": int(K), "variants": {}}
    for name, lanes in (("f32_sequential", 1), ("f32_simd8", 8), ("f32_simd16", 16)):
        set_diff = order_diff = inconclusive = 0
        max_abs = max_rel = 0.0
        for qi in range(nq):
            rr = rows64[qi]
            cand = fetch_rows(rr)
            d32 = o.l2_f32acc_many(cand, queries[qi], lanes)
            d64 = dist64[qi]
            err = float(np.abs(d32.astype(np.float64) - d64.astype(np.float64)).max())
            max_abs = max(max_abs, err)
            max_rel = max(max_rel, float((np.abs(d32.astype(np.float64) - d64) / np.maximum(d64, 1e-30)).max()))
            if not (float(d64[K - 1]) - float(d64[k - 1]) > 2.0 * err):
                inconclusive += 1
            order32 = np.lexsort((rr, d32))[:k]          # (distance asc, row id asc)
            top32 = rr[order32]
            if set(top32.tolist()) != set(rr[:k].tolist()):
                set_diff += 1
            if not np.array_equal(top32, rr[:k]):
                order_diff += 1
        out["variants"][name] = {"top_k_sets_that_differ": set_diff, "top_k_orders_that_differ": order_diff,
                                 "queries_not_conclusive": inconclusive, "max_abs_distance_difference": max_abs,
                                 "max_rel_distance_difference": max_rel}
    return out
