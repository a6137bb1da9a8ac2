"""Shared test inputs: golden fixture loading and the reference tests' own data recipes."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def pattern(n):  # makePatternData, tests/unit/chunking/chunking_test.cpp:55-62 (reference)
    i = np.arange(n, dtype=np.uint64)
    return ((i * np.uint64(1315423911) + np.uint64(0x9E3779B9)) & np.uint64(0xFF)).astype(np.uint8)


def gen_input(spec, oracle=None):
    kind, n = spec["kind"], spec["n"]
    if kind == "random":
        return np.random.default_rng(spec["seed"]).integers(0, 256, n, dtype=np.uint8)
    if kind == "pattern":
        return pattern(n)
    if kind == "zeros":
        return np.zeros(n, np.uint8)
    if kind == "const":
        return np.full(n, spec["value"], np.uint8)
    if kind == "philox":
        return oracle.synth_bytes(spec["seed"], spec.get("blob", 0), 0, n)
    raise ValueError(kind)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)["cases"]


def fixture_embedding(dim, seed):
    """SqliteVecBackendFixture::createEmbedding (reference tests/unit/vector/
    sqlite_vec_backend_comprehensive_catch2_test.cpp:84-101): mt19937(seed*1000), U(-1,1), fp32
    normalise.  numpy's MT19937 with the same 32-bit seed yields the same raw 32-bit stream;
    libstdc++'s uniform_real_distribution<float> maps one draw to (x / 2^32) * 2 - 1 in float."""
    bg = np.random.MT19937()
    # seed exactly like std::mt19937(uint32 seed)
    st = np.zeros(624, dtype=np.uint32)
    st[0] = np.uint32(seed * 1000)
    for i in range(1, 624):
        st[i] = np.uint32((1812433253 * (int(st[i - 1]) ^ (int(st[i - 1]) >> 30)) + i) & 0xFFFFFFFF)
    bg.state = {"bit_generator": "MT19937", "state": {"key": st, "pos": 624}}
    raw = bg.random_raw(dim).astype(np.uint32)
    u = (raw.astype(np.float32) * np.float32(2.3283064365386963e-10))  # generate_canonical<float,24>
    u = np.minimum(u, np.nextafter(np.float32(1.0), np.float32(0.0)))
    v = u * np.float32(2.0) + np.float32(-1.0)
    nrm = np.float32(0.0)
    for x in v:
        nrm = np.float32(nrm + x * x)
    nrm = np.sqrt(nrm, dtype=np.float32)
    return (v / nrm).astype(np.float32) if nrm > 0 else v.astype(np.float32)


def string_ranks(ids):
    """tie_rank[row] = rank of the row's chunk_id in lexicographic (byte) order
    (the comparator of sqlite_vec_backend.cpp:4218-4223 compares std::string chunk ids)."""
    order = sorted(range(len(ids)), key=lambda i: ids[i].encode())
    rank = np.zeros(len(ids), np.uint32)
    for r, i in enumerate(order):
        rank[i] = r
    inv = np.array(order, np.uint32)
    return rank, inv


# ---- tests/golden/scan.json: inputs are regenerated from the case's recipe, only outputs are stored -------------------
def _bits_rows(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def golden_scan_matrix(oracle, spec):
    """{"kind": "mt19937" | "philox" | "normal" | "bits", ...} -> float32 [n][dim]; "overrides": {row: [u32 bits]} replaces rows
    (NaN / zero / tiny-norm / huge rows), "repeat": [[dst_lo, dst_hi, src]] makes runs of identical rows (ties)."""
    kind = spec["kind"]
    if kind == "mt19937":            # the reference's own recipe, vector_backend_engine_compare.cpp:83-107
        m = oracle.mt19937_rows(spec["seed"], spec.get("skip", 0), spec["n"], spec["dim"])
    elif kind == "philox":           # the synthetic-embedding recipe of bench.py and the GPU tests
        m = oracle.synth_rows(spec["seed"], spec.get("row0", 0), spec["n"], spec["dim"])
    elif kind == "normal":
        m = np.random.default_rng(spec["seed"]).standard_normal((spec["n"], spec["dim"])).astype(np.float32)
    elif kind == "bits":
        m = _bits_rows(spec["rows"])
    elif kind == "perm_offsets":     # rows = base[0] + a permutation of ONE offset vector: equal distances to base[0] in exact
        base = golden_scan_matrix(oracle, spec["base"])[0]      # arithmetic, different fp32 roundings per summation order
        rng = np.random.default_rng(spec["seed"])
        e = (rng.standard_normal(spec["dim"]) * spec.get("scale", 1.0)).astype(np.float32)
        m = np.stack([base + e[rng.permutation(spec["dim"])] for _ in range(spec["n"])]).astype(np.float32)
    else:
        raise ValueError(kind)
    m = np.array(m, np.float32, copy=True)
    for lo, hi, src in spec.get("repeat", []):
        m[lo:hi] = m[src]
    for r, bits in spec.get("overrides", {}).items():
        m[int(r)] = _bits_rows(bits)
    return m


def golden_scan_ids(case, n):
    """Chunk ids of the case's rows: explicit, a seeded shuffle of zero-padded numbers, or None (ordinals: chunk-id order
    == row order)."""
    ids = case.get("chunk_ids")
    if isinstance(ids, dict):
        perm = np.random.default_rng(ids["shuffle_seed"]).permutation(n)
        return [ids.get("prefix", "id_") + "%06d" % v for v in perm]
    return ids


def golden_scan_inputs(oracle, case):
    """(corpus, queries, tie_rank | None, allow | None) of one golden case."""
    corpus = golden_scan_matrix(oracle, case["corpus"])
    queries = golden_scan_matrix(oracle, case["queries"])
    ids = golden_scan_ids(case, corpus.shape[0])
    tie_rank = string_ranks(ids)[0] if ids is not None else None
    allow = None
    if case.get("allow_every"):
        allow = (np.arange(corpus.shape[0]) % case["allow_every"] == 0).astype(np.uint8)
    return corpus, queries, tie_rank, allow
