"""Generates tests/golden/scan_l2.json from the REFERENCE'S OWN vec0SearchUnlocked (round 6).

    python tests/golden/make_scan_l2_golden.py        (dev container: needs /root/reference; builds oracle/_ref)

Inputs are regenerated from recipes (tests/_cases.py golden_scan_inputs); every case is inserted into an in-memory `vectors`
table, indexed by the reference's rebuildVec0DimUnlocked and searched by its vec0SearchUnlocked as compiled into
oracle/_ref/libyams_scan_ref.so (cut verbatim by oracle/gen_scan_ref.py).  The `vec0` virtual table is the harness's
(the dependency that owns it is absent from the checkout): its distance function is plugged per definition — "f64", "f32",
"f32x8", "f32x16" and the fused forms — so every case carries one expected result PER DEFINITION.  What the file pins on
the GPU box is the post-processing the reference owns (k nearest then the cosine threshold, rowid order among equal
distances whatever the chunk ids, the candidate restriction) on top of a distance definition the host calibrates; the
arithmetic itself stays unpinned.  Only OUTPUTS are stored: row ordinals in result order, the bits of relevance_score."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _cases  # noqa: E402
import _oracle  # noqa: E402

DEFINITIONS = {"f64": None, "f32": 1, "f32x8": 8, "f32x16": 16, "f32_fma": -1, "f32x8_fma": -8, "f32x16_fma": -16}

CASES = [
    {"name": "l2_mt19937_3000_x384_top10", "k": 10, "threshold": -1.0,
     "corpus": {"kind": "mt19937", "seed": 42, "n": 3000, "dim": 384}, "queries": {"kind": "mt19937", "seed": 42, "skip": 3000, "n": 8, "dim": 384}},
    {"name": "l2_mt19937_3000_x384_top100_threshold", "k": 100, "threshold": 0.12,      # k nearest FIRST, then the threshold: fewer than k
     "corpus": {"kind": "mt19937", "seed": 42, "n": 3000, "dim": 384}, "queries": {"kind": "mt19937", "seed": 42, "skip": 3000, "n": 6, "dim": 384}},
    {"name": "l2_philox_20000_x256_top100", "k": 100, "threshold": -1.0,                 # large enough for the MFMA filter tiers
     "corpus": {"kind": "philox", "seed": 17, "n": 20000, "dim": 256}, "queries": {"kind": "philox", "seed": 17, "row0": 1 << 40, "n": 6, "dim": 256}},
    {"name": "l2_philox_6000_x768_top50_threshold", "k": 50, "threshold": 0.02,
     "corpus": {"kind": "philox", "seed": 42, "n": 6000, "dim": 768}, "queries": {"kind": "philox", "seed": 42, "row0": 1 << 40, "n": 5, "dim": 768}},
    {"name": "l2_k_above_n", "k": 100, "threshold": -1.0,
     "corpus": {"kind": "philox", "seed": 3, "n": 7, "dim": 16}, "queries": {"kind": "philox", "seed": 3, "row0": 100, "n": 3, "dim": 16}},
    # equal distances under shuffled chunk ids: rowid order (the cut falls inside a run of duplicates for k = 7, 19)
    *[{"name": f"l2_ties_in_rowid_order_k{k}", "k": k, "threshold": -1.0, "chunk_ids": {"shuffle_seed": 5},
       "corpus": {"kind": "normal", "seed": 11, "n": 60, "dim": 12, "repeat": [[0, 20, 0], [20, 35, 20], [35, 50, 35]]},
       "queries": {"kind": "normal", "seed": 12, "n": 4, "dim": 12}} for k in (1, 7, 19, 60)],
    {"name": "l2_ties_philox_8000_x256_top40", "k": 40, "threshold": -1.0, "chunk_ids": {"shuffle_seed": 8, "prefix": "k"},
     "corpus": {"kind": "philox", "seed": 35, "n": 8000, "dim": 256, "repeat": [[16, 40, 16], [5000, 5030, 16]]},
     "queries": {"kind": "philox", "seed": 35, "row0": 16, "n": 3, "dim": 256}},        # the queries ARE rows 16.. (distance 0 to 54 rows)
    # rows at (nearly) the SAME distance from the query — the query plus permutations of one offset vector: the fp64 definition
    # sees mostly exact ties (rowid order), every fp32 definition its own rounding of the same sum: the one case of this file
    # whose expected result DEPENDS on the definition the host calibrates to
    {"name": "l2_definition_sensitive_permuted_offsets", "k": 12, "threshold": -1.0,
     "corpus": {"kind": "perm_offsets", "seed": 77, "n": 48, "dim": 256, "scale": 3.0, "base": {"kind": "normal", "seed": 78, "n": 1, "dim": 256}},
     "queries": {"kind": "normal", "seed": 78, "n": 1, "dim": 256}},
    # candidate rowids (every fourth row), no threshold: KNN among the candidates
    {"name": "l2_candidates_philox_5000_x256_top40", "k": 40, "threshold": -1.0, "allow_every": 4,
     "corpus": {"kind": "philox", "seed": 36, "n": 5000, "dim": 256, "repeat": [[16, 24, 16]]},
     "queries": {"kind": "philox", "seed": 36, "row0": 1 << 40, "n": 4, "dim": 256}},
]


def main():
    o = _oracle.oracle()
    t0 = _oracle.scan_ref()
    if t0 is None or not t0.has_vec0:
        raise SystemExit("oracle/_ref/libyams_scan_ref.so (with the vec0 doors) is missing: run `make -C oracle` where /root/reference exists")
    t0.close()
    out_cases = []
    for case in CASES:
        corpus, queries, _, allow = _cases.golden_scan_inputs(o, case)
        ids = _cases.golden_scan_ids(case, corpus.shape[0])
        t = _oracle.scan_ref()
        t.insert_rows(corpus, chunk_ids=ids)
        cand = None
        if allow is not None:
            cand = [t.rowid_of(int(r)) for r in np.flatnonzero(allow)]
        expected = {}
        for name, lanes in DEFINITIONS.items():
            t.vec0_set_distance(lanes)
            t.vec0_rebuild(corpus.shape[1])
            per_q = []
            for q in queries:
                r = t.vec0_search(q, case["k"], case["threshold"], candidate_rowids=cand)
                assert not isinstance(r, int), r
                per_q.append({"rows": [int(x) for x in r[0]], "score_bits": [int(x) for x in r[1].view(np.uint32)]})
            expected[name] = per_q
        t.close()
        c = dict(case)
        # definitions whose results equal f64's are stored once
        c["expected"] = {"f64": expected["f64"]}
        c["same_as_f64"] = [n_ for n_ in DEFINITIONS if n_ != "f64" and expected[n_] == expected["f64"]]
        for n_ in DEFINITIONS:
            if n_ != "f64" and n_ not in c["same_as_f64"]:
                c["expected"][n_] = expected[n_]
        out_cases.append(c)
    spans = open(os.path.join(_oracle.ORACLE_DIR, "_ref", "scan_ref_spans.txt")).read().split("\n")
    doc = {"generator": "tests/golden/make_scan_l2_golden.py",
           "what": "outputs of the reference's own vec0SearchUnlocked (compiled from /root/reference by oracle/Makefile: "
                   "_ref/libyams_scan_ref.so) over the harness's vec0 module, one result per plugged distance definition; rows = "
                   "ordinals in result order, score_bits = the fp32 bits of relevance_score (the cosine re-score)",
           "definitions": list(DEFINITIONS), "reference_spans": [s for s in spans if s and ("ec0" in s or "Rowid" in s or "Cosine" in s or "Guard" in s)],
           "cases": out_cases}
    path = os.path.join(HERE, "scan_l2.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes,", len(out_cases), "cases,",
          sum(len(v) for c in out_cases for v in c["expected"].values()), "stored results")


if __name__ == "__main__":
    main()
