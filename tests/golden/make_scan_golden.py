"""Generates tests/golden/scan.json from the REFERENCE'S OWN exact-scan loop.

    python tests/golden/make_scan_golden.py        (dev container: needs /root/reference; builds oracle/_ref)

Every case's rows and queries are regenerated from a recipe (tests/_cases.py golden_scan_inputs); they are inserted into
an in-memory `vectors` table and searched by SqliteVecBackend::Impl::bruteForceSearchUnlocked as compiled into
oracle/_ref/libyams_scan_ref.so (cut verbatim from /root/reference by oracle/gen_scan_ref.py).  Only the OUTPUTS are
stored: row ordinals in result order and the bits of every returned score.  The file pins oracle/yams_oracle.c
(tests/test_scan_ref_pin.py) and the HIP path (tests/test_scan_gpu.py) on the GPU box, where /root/reference does not
exist."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _cases  # noqa: E402
import _oracle  # noqa: E402


def f32bits(*vals):
    return [int(x) for x in np.array(vals, np.float32).view(np.uint32)]


L = float(np.finfo(np.float32).max / 4)
SKIP_ROWS = [f32bits(1, 0, 0, 0, 0, 0, 0, 0), f32bits(0, 0, 0, 0, 0, 0, 0, 0), f32bits(np.nan, 1, 0, 0, 0, 0, 0, 0),
             f32bits(1e-7, 0, 0, 0, 0, 0, 0, 0), f32bits(2, 0, 0, 0, 0, 0, 0, 0), f32bits(L, -L, L, -L, L, -L, L, -L),
             f32bits(-L, L, -L, L, -L, L, -L, L), f32bits(0, np.inf, 0, 0, 0, 0, 0, 0), f32bits(1.1e-6, 0, 0, 0, 0, 0, 0, 0),
             f32bits(0.5, 0.5, 0, 0, 0, 0, 0, 0), f32bits(3e-6, 0, 0, 0, 0, 0, 0, 0)]
SKIP_QUERIES = [SKIP_ROWS[0], SKIP_ROWS[5], SKIP_ROWS[9], f32bits(1, 1, 1, 1, 1, 1, 1, 1), f32bits(L, L, L, L, L, L, L, L)]
BAD_QUERIES = [f32bits(0, 0, 0, 0, 0, 0, 0, 0), f32bits(np.nan, 0, 0, 0, 0, 0, 0, 0), f32bits(np.inf, 0, 0, 0, 0, 0, 0, 0),
               f32bits(9e-6, 0, 0, 0, 0, 0, 0, 0), f32bits(1.1e-5, 0, 0, 0, 0, 0, 0, 0)]

CASES = [
    # BASELINE config 1 on the reference's own embedding recipe
    {"name": "config1_mt19937_10k_x384_top10", "path": "fast", "k": 10, "threshold": -1.0,
     "corpus": {"kind": "mt19937", "seed": 42, "n": 10000, "dim": 384}, "queries": {"kind": "mt19937", "seed": 42, "skip": 10000, "n": 16, "dim": 384}},
    {"name": "config1_mt19937_threshold_0p1", "path": "fast", "k": 10, "threshold": 0.1,
     "corpus": {"kind": "mt19937", "seed": 42, "n": 10000, "dim": 384}, "queries": {"kind": "mt19937", "seed": 42, "skip": 10000, "n": 8, "dim": 384}},
    # the bench's row family (Philox, dim 768 / 384), top-100
    {"name": "philox_6000_x768_top100", "path": "fast", "k": 100, "threshold": -1.0,
     "corpus": {"kind": "philox", "seed": 42, "n": 6000, "dim": 768}, "queries": {"kind": "philox", "seed": 42, "row0": 1 << 40, "n": 8, "dim": 768}},
    {"name": "philox_20000_x384_top100_threshold", "path": "fast", "k": 100, "threshold": 0.04,
     "corpus": {"kind": "philox", "seed": 7, "n": 20000, "dim": 384}, "queries": {"kind": "philox", "seed": 7, "row0": 1 << 40, "n": 6, "dim": 384}},
    # k above the row count, k = 1
    {"name": "k_above_n", "path": "fast", "k": 100, "threshold": -1.0,
     "corpus": {"kind": "philox", "seed": 3, "n": 7, "dim": 16}, "queries": {"kind": "philox", "seed": 3, "row0": 100, "n": 3, "dim": 16}},
    {"name": "top1", "path": "fast", "k": 1, "threshold": -1.0,
     "corpus": {"kind": "normal", "seed": 9, "n": 3000, "dim": 64}, "queries": {"kind": "normal", "seed": 10, "n": 8, "dim": 64}},
    # equal scores ordered by chunk_id under shuffled string ids, heap at several fill levels
    *[{"name": f"ties_by_chunk_id_k{k}", "path": "fast", "k": k, "threshold": -1.0, "chunk_ids": {"shuffle_seed": 5},
       "corpus": {"kind": "normal", "seed": 11, "n": 60, "dim": 12, "repeat": [[0, 20, 0], [20, 35, 20], [35, 50, 35]]},
       "queries": {"kind": "normal", "seed": 12, "n": 4, "dim": 12}} for k in (1, 7, 19, 60)],
    # rows the loop skips, +-FLT_MAX/4 rows, the 1e-12 bound
    {"name": "skips_and_extremes", "path": "fast", "k": 20, "threshold": -1.0,
     "corpus": {"kind": "bits", "rows": SKIP_ROWS}, "queries": {"kind": "bits", "rows": SKIP_QUERIES}},
    {"name": "skips_and_extremes_threshold_0p5", "path": "fast", "k": 20, "threshold": 0.5,
     "corpus": {"kind": "bits", "rows": SKIP_ROWS}, "queries": {"kind": "bits", "rows": SKIP_QUERIES}},
    # invalid queries (InvalidArgument), the last one is just valid
    {"name": "invalid_queries", "path": "fast", "k": 3, "threshold": -1.0,
     "corpus": {"kind": "bits", "rows": SKIP_ROWS}, "queries": {"kind": "bits", "rows": BAD_QUERIES}},
    # the metadata-filter (record) path: every third row matches; the 1e-10 rule; TopK and AllMatching
    {"name": "record_path_top10", "path": "record", "k": 10, "threshold": -1.0, "allow_every": 3, "chunk_ids": {"shuffle_seed": 6, "prefix": "m"},
     "corpus": {"kind": "normal", "seed": 21, "n": 600, "dim": 24, "repeat": [[99, 111, 99]],
                "overrides": {"3": f32bits(*([0.0] * 24)), "6": f32bits(2e-6, *([0.0] * 23)), "9": f32bits(np.nan, *([1.0] * 23))}},
     "queries": {"kind": "normal", "seed": 22, "n": 5, "dim": 24}},
    {"name": "record_path_all_matching_threshold", "path": "record", "k": 0, "threshold": 0.05, "all_matching": True, "allow_every": 3,
     "chunk_ids": {"shuffle_seed": 6, "prefix": "m"},
     "corpus": {"kind": "normal", "seed": 21, "n": 600, "dim": 24, "repeat": [[99, 111, 99]],
                "overrides": {"3": f32bits(*([0.0] * 24)), "6": f32bits(2e-6, *([0.0] * 23)), "9": f32bits(np.nan, *([1.0] * 23))}},
     "queries": {"kind": "normal", "seed": 22, "n": 3, "dim": 24}},
    # candidate restriction on the FAST path (candidate_hashes pushed into SQL, :4151-4195): every fourth row's document is named
    {"name": "candidates_fast_path_philox_5000_x256_top40", "path": "fast", "k": 40, "threshold": -1.0, "allow_every": 4,
     "chunk_ids": {"shuffle_seed": 8, "prefix": "k"},
     "corpus": {"kind": "philox", "seed": 35, "n": 5000, "dim": 256, "repeat": [[16, 24, 16]]},
     "queries": {"kind": "philox", "seed": 35, "row0": 1 << 40, "n": 5, "dim": 256}},
    {"name": "record_path_philox_4000_x256_top50", "path": "record", "k": 50, "threshold": -1.0, "allow_every": 2,
     "corpus": {"kind": "philox", "seed": 31, "n": 4000, "dim": 256}, "queries": {"kind": "philox", "seed": 31, "row0": 1 << 40, "n": 4, "dim": 256}},
]


def main():
    o = _oracle.oracle()
    if _oracle.scan_ref() is None:
        raise SystemExit("oracle/_ref/libyams_scan_ref.so is missing: run `make -C oracle` where /root/reference exists")
    out_cases = []
    for case in CASES:
        corpus, queries, _, allow = _cases.golden_scan_inputs(o, case)
        ids = _cases.golden_scan_ids(case, corpus.shape[0])
        t = _oracle.scan_ref()
        cands = None
        if case["path"] == "record":
            for i in range(corpus.shape[0]):
                t.insert_raw(ids[i] if ids else "c%018d" % i, corpus[i].tobytes(), corpus.shape[1],
                             metadata={"tag": "a" if allow[i] else "b"})
        elif allow is not None:     # fast path behind a candidate set: the allowed rows live in the named document
            t.insert_rows(corpus, chunk_ids=ids, document_hashes=["named" if a else "other" for a in allow])
            cands = {"named"}
        else:
            t.insert_rows(corpus, chunk_ids=ids)
        expected = []
        for q in queries:
            r = t.search(q, case["k"], case["threshold"], metadata_filters={"tag": "a"} if case["path"] == "record" else None,
                         all_matching=case.get("all_matching", False), candidate_hashes=cands)
            if isinstance(r, int):
                assert r == t.invalid_argument, r
                expected.append({"error": "InvalidArgument"})
            else:
                expected.append({"rows": [int(x) for x in r[0]], "score_bits": [int(x) for x in r[1].view(np.uint32)],
                                 "rows_visited": int(r[2]["rows_visited"]), "evaluations": int(r[2]["exact_distance_evaluations"])})
        t.close()
        c = dict(case)
        c["expected"] = expected
        out_cases.append(c)
    spans = open(os.path.join(_oracle.ORACLE_DIR, "_ref", "scan_ref_spans.txt")).read().split("\n")
    doc = {"generator": "tests/golden/make_scan_golden.py",
           "what": "outputs of the reference's own bruteForceSearchUnlocked (compiled from /root/reference by oracle/Makefile: "
                   "_ref/libyams_scan_ref.so) on regenerable inputs; rows = ordinals in result order, score_bits = the fp32 bits of "
                   "relevance_score",
           "reference_spans": [s for s in spans if s], "cases": out_cases}
    path = os.path.join(HERE, "scan.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes,", len(out_cases), "cases,", sum(len(c["expected"]) for c in out_cases), "queries")


if __name__ == "__main__":
    main()
