"""Generates tests/golden/*.json from the REFERENCE's own translation units.

Run in the dev container (needs /root/reference; builds oracle/_ref via oracle/Makefile):
    python tests/golden/make_golden.py
The fixtures pin oracle/yams_oracle.c and the HIP path on the GPU box, where /root/reference
does not exist.  Inputs are regenerated from seeds, only outputs are stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle  # noqa: E402


def pattern(n):  # makePatternData, tests/unit/chunking/chunking_test.cpp:55-62
    i = np.arange(n, dtype=np.uint64)
    return ((i * np.uint64(1315423911) + np.uint64(0x9E3779B9)) & np.uint64(0xFF)).astype(np.uint8)


def gen_input(spec):
    kind = spec["kind"]
    n = spec["n"]
    if kind == "random":
        return np.random.default_rng(spec["seed"]).integers(0, 256, n, dtype=np.uint8)
    if kind == "pattern":
        return pattern(n)
    if kind == "zeros":
        return np.zeros(n, np.uint8)
    if kind == "const":
        return np.full(n, spec["value"], np.uint8)
    if kind == "philox":
        return _oracle.oracle().synth_bytes(spec["seed"], spec.get("blob", 0), 0, n)
    raise ValueError(kind)


CDC_CASES = [
    # (input spec, config overrides)
    ({"kind": "random", "seed": 11, "n": 4 << 20}, {}),
    ({"kind": "random", "seed": 12, "n": (1 << 20) + 12345}, {"min_size": 4096, "max_size": 65536}),
    ({"kind": "random", "seed": 13, "n": 300000}, {"min_size": 2048, "max_size": 16384, "mask": 0x7FF}),
    ({"kind": "random", "seed": 14, "n": 100000}, {"min_size": 64, "max_size": 1024, "mask": 0x3F, "window": 16}),
    ({"kind": "random", "seed": 15, "n": 70000}, {"min_size": 1000, "max_size": 5000, "mask": 0xFFFFF}),
    ({"kind": "pattern", "n": 256 * 1024 + 777}, {"min_size": 2048, "max_size": 65536}),
    ({"kind": "pattern", "n": 256 * 1024 + 777}, {}),
    ({"kind": "zeros", "n": 3 << 20}, {}),
    ({"kind": "const", "value": 0x42, "n": 200000}, {"min_size": 4096, "max_size": 32768}),
    ({"kind": "philox", "seed": 42, "blob": 3, "n": 2 << 20}, {}),
    ({"kind": "random", "seed": 16, "n": 47}, {}),
    ({"kind": "random", "seed": 17, "n": 16384}, {}),
    ({"kind": "random", "seed": 18, "n": 16385}, {}),
    ({"kind": "random", "seed": 19, "n": 1}, {"min_size": 1, "max_size": 4, "mask": 1}),
]


def main():
    r = _oracle.ref()
    assert r is not None, "needs oracle/_ref (reference sources) — run in the dev container"
    sha = []
    for msg in [b"", b"abc", b"Hello World"]:  # tests/unit/crypto/crypto_test.cpp:92-99
        sha.append({"ascii": msg.decode(), "hex": r.sha256_hex(msg)})
    rng = np.random.default_rng(2024)
    for n in [1, 17, 55, 56, 57, 63, 64, 65, 119, 120, 121, 127, 128, 129, 4096, 65537]:
        seed = int(rng.integers(1 << 30))
        data = np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)
        sha.append({"seed": seed, "n": n, "hex": r.sha256_hex(data)})
    cdc = []
    for spec, cfg in CDC_CASES:
        data = gen_input(spec)
        for mode in ("rabin", "streaming"):
            off, sz, hashes = r.chunks(data, mode, with_hashes=True, **cfg)
            cdc.append({"input": spec, "config": cfg, "mode": mode,
                        "offsets": [int(x) for x in off], "sizes": [int(x) for x in sz],
                        # keep the fixture small: first/last few chunk hashes + a digest of all
                        "hash_head": hashes[:3], "hash_tail": hashes[-3:],
                        "hash_of_hashes": r.sha256_hex("".join(hashes).encode())})
    with open(os.path.join(HERE, "sha256.json"), "w") as f:
        json.dump({"source": "reference src/crypto/sha256_hasher.cpp via oracle/_ref", "cases": sha}, f, indent=1)
    with open(os.path.join(HERE, "cdc.json"), "w") as f:
        json.dump({"source": "reference src/chunking/{rabin,streaming}_chunker.cpp via oracle/_ref",
                   "cases": cdc}, f)
    # the reference's synthetic-embedding recipe (vector_backend_engine_compare.cpp:83-107) drawn
    # with the real std::mt19937 / std::uniform_real_distribution<float> (oracle/ref_wrap.cpp)
    import hashlib
    small = r.mt19937_rows(7, 5, 6)
    c1 = r.mt19937_rows(42, 10_000 + 16, 384)      # BASELINE config 1: corpus, then the queries
    mt = {"source": "std::mt19937 + std::uniform_real_distribution<float>(-1,1) + fp32 normalise, libstdc++ "
                    "of the dev container, via oracle/_ref (ref_mt19937_rows)",
          "small": {"seed": 7, "count": 5, "dim": 6, "values_u32": [int(x) for x in small.view(np.uint32).ravel()]},
          "config1": {"seed": 42, "count": 10_016, "dim": 384,
                      "sha256_of_float32_bytes": hashlib.sha256(c1.tobytes()).hexdigest(),
                      "first_row_head_u32": [int(x) for x in c1[0, :8].view(np.uint32)],
                      "first_query_head_u32": [int(x) for x in c1[10_000, :8].view(np.uint32)]}}
    with open(os.path.join(HERE, "mt_recipe.json"), "w") as f:
        json.dump(mt, f, indent=1)
    print("wrote", len(sha), "sha cases and", len(cdc), "cdc cases + the mt19937 recipe")


if __name__ == "__main__":
    main()
